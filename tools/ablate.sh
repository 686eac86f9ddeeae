for a in 1 2 3 0; do HOPE_DEBUG_ABLATE=$a timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames 256 > gpurun_out/abl_$a.log 2>&1; done
