for tag in w2d4 w2d2; do
  MCMCPP_HIP_LIB=$PWD/mcmcpp_amd/libmcmcpp_hip_$tag.so timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or fresh or matrix_core" 2>&1 | tail -1
done
bash tools/ab_libs.sh base w2d4 w2d2 base
for tag in base w2d4 w2d2; do
  lib=$PWD/mcmcpp_amd/libmcmcpp_hip_$tag.so; [ "$tag" = base ] && lib=$PWD/mcmcpp_amd/libmcmcpp_hip.so
  MCMCPP_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-live-counters --steps 6 --warmup 2 --no-cpu-baseline --no-chain --calc iso 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('iso', '$tag', '%.3e %.2f us'%(d['value'], d['roofline']['avg_launch_us']))"
done
