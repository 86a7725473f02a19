cd $GRAFT_REPO_ROOT
for w in 7 8; do
  echo "== pipe waves=$w"; RT_HIP_LIB=$GRAFT_REPO_ROOT/raytracing-project_amd/csrc/variants/librt_hip_p$w.so timeout -k 10 60 python3 tools/prof_frame.py final 3840 2160 4 3 2>&1 | grep -v materials | tail -1 || exit 1
  RT_HIP_LIB=$GRAFT_REPO_ROOT/raytracing-project_amd/csrc/variants/librt_hip_p$w.so WORLD=8 RT_WF_REGION=64 timeout -k 10 60 python3 tools/prof_frame.py final 3840 2160 4 3 2>&1 | grep -v materials | tail -1 || exit 1
done
