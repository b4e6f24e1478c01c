cd /root/repo
timeout 400 python tools/train_demo.py 6000 1e-5 2000000 3 700 disc_step_bias=5 n_steps_per_image=3 > gpurun_out/r02_train_curve_disc_step_bias5.json 2> /tmp/err.log; tail -4 /tmp/err.log | cut -c1-300
