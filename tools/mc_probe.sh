cd $GRAFT_REPO_ROOT
for dw in 1 0; do
 echo "== MCMCPP_HIP_MC_DRAW_WAVES=$dw"
 MCMCPP_HIP_MC_DRAW_WAVES=$dw python tools/time_config.py 16384 32 dense f64 2000 8
 MCMCPP_HIP_MC_DRAW_WAVES=$dw python tools/time_config.py 65536 32 dense f64 500
 MCMCPP_HIP_MC_DRAW_WAVES=$dw python tools/time_config.py 131072 32 dense f64 500
 MCMCPP_HIP_MC_DRAW_WAVES=$dw python tools/time_config.py 262144 32 dense f64 200
 MCMCPP_HIP_MC_DRAW_WAVES=$dw python tools/time_config.py 131072 32 dense f32 500
done
