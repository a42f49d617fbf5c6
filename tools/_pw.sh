for W in 1 2 3; do MGL_PICK_WAVES=$W timeout -k 10 200 python bench.py --no-cpu > gpurun_out/b_c2_w$W.json || exit 1; done
timeout -k 10 300 python bench.py --config c3 --steps 200 --warmup 20 --no-cpu > gpurun_out/b_c3.json
