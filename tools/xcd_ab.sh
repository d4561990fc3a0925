for X in 0 1; do
  echo "== GRK_AMD_DWT_XCD=$X"; export GRK_AMD_DWT_XCD=$X
  timeout 200 python tools/k3_time.py head 2>&1 | grep -v amdgpu | tail -4
  bash tools/pmc_one.sh "FETCH_SIZE" 2>&1 | grep -E "dwt0|dwtN"
done
