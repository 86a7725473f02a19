set -e
for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1"; do
for S in 64 128 256 512; do
  echo "== $cfg S=$S"
  RT_WF_REGION=$S timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
done; done
