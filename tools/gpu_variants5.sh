pat=$1; shift
for rep in 1 2 3 4 5; do for v in "$@"; do
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/$v/librtm3d_hip.so "$pat" 2>/dev/null || exit 1
done; done
