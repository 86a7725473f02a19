cd $GRAFT_REPO_ROOT
for cfg in "4 3" "2 3" "2 5" "1 5" "1 8" "3 4"; do set -- $cfg
  echo "== leaf=$1 ci=$2"; RT_KD_LEAF=$1 RT_KD_CI=$2 timeout -k 10 60 python3 tools/prof_frame.py final 3840 2160 4 3 2>&1 | grep -v materials | tail -1 || exit 1
done
