cd $GRAFT_REPO_ROOT
for w in 8 6 5; do for kb in 40 24 12; do
  echo "== waves=$w lds_kb=$kb"; RT_HIP_LIB=$GRAFT_REPO_ROOT/raytracing-project_amd/csrc/variants/librt_hip_w$w.so RT_WF_LDS_KB=$kb timeout -k 10 60 python3 tools/prof_frame.py final 3840 2160 4 3 2>&1 | grep -v materials | tail -1 || exit 1
done; done
