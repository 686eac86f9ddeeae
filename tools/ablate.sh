for a in 10 20 30 0; do HOPE_DEBUG_ABLATE=$a timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames 1024 > gpurun_out/abl_$a.log 2>&1; done
