cd $GRAFT_REPO_ROOT
for dt in f64 f32; do
 echo "== $dt default"; python tools/time_config.py 131072 64 iso $dt
 echo "== $dt no draw wave"; MCMCPP_HIP_NO_DRAW_WAVE=1 python tools/time_config.py 131072 64 iso $dt
 for p in 1 2 4; do echo "== $dt passes $p"; MCMCPP_HIP_PASSES=$p python tools/time_config.py 131072 64 iso $dt; done
done
echo "== f32 131072x32"; python tools/time_config.py 131072 32 iso f32
echo "== f64 131072x32"; python tools/time_config.py 131072 32 iso f64
echo "== f32 C2 dense"; python tools/time_config.py 16384 32 dense f32 2000
echo "== f32 C2 dense half-step"; MCMCPP_HIP_FULL_STEP=0 python tools/time_config.py 16384 32 dense f32 2000
echo "== f32 C2 iso"; python tools/time_config.py 16384 32 iso f32 2000
echo "== f64 C2 iso"; python tools/time_config.py 16384 32 iso f64 2000
