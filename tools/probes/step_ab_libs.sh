# same-box A/B of two library builds and the relu-mask switch: bash tools/probes/step_ab_libs.sh [batch]
R=${GRAFT_REPO_ROOT:-.}; OLD=$R/predict_pv_yield_amd/lib/libpvyield_old.so
for i in 1 2 3; do
  PV_YIELD_LIB=$OLD python3 $R/tools/probes/step_time.py $1 2>/dev/null
  python3 $R/tools/probes/step_time.py $1 2>/dev/null
  PV_AB_SET=functional.USE_RELU_MASKS=1 python3 $R/tools/probes/step_time.py $1 2>/dev/null
done
