# same-box A/B of two builds of the library on the advection pipeline's stage times: bash tools/probes/flow_ab_libs.sh
R=${GRAFT_REPO_ROOT:-.}; L=$R/predict_pv_yield_amd/lib
for i in 1 2 3; do
  for v in base var; do
    echo "== $v"; PV_YIELD_LIB=$L/libpvyield_$v.so python3 $R/tools/time_flow_stages.py 32 2>/dev/null | grep -E "polyexp|iterations|pipeline|total" | head -8
  done
done
