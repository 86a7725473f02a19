set -e
for leaf in 4 6 8; do
  echo "== leaf $leaf"
  for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1" "lego 3840 2160 4" "final 3840 2160 4"; do
    RT_KD_LEAF=$leaf timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
  done
  RT_KD_LEAF=$leaf WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 6 | grep "no-stats"
done
