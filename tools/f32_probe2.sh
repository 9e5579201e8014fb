cd $GRAFT_REPO_ROOT
for lib in "" notie; do
 [ -n "$lib" ] && export MCMCPP_HIP_LIB=$PWD/mcmcpp_amd/libmcmcpp_hip_$lib.so
 echo "== lib ${lib:-base}"
 python tools/time_config.py 131072 32 iso f32
 python tools/time_config.py 131072 64 iso f32
 python tools/time_config.py 65536 32 rosenbrock f32
 python tools/time_config.py 65536 32 rosenbrock f64
 python tools/time_config.py 16384 32 dense f32 2000
 python tools/time_config.py 131072 64 iso f64
done
