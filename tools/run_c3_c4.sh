# BASELINE configs 3 and 4 at full size through the drivers (development aid; the assertions live in tests/test_gpu_fullsize_configs.py)
cd $GRAFT_REPO_ROOT/nonlinpdes-gpsolver_amd
( time python main_Burgers1d.py --N_domain 2000 --N_boundary 400 --GNsteps 8 --show_figure "" ) 2>&1 | tail -25
( time python main_DarcyFlow2d.py --N_domain 1600 --N_boundary 200 --N_data 60 --noise_level 1e-3 --show_figure "" ) 2>&1 | tail -25
