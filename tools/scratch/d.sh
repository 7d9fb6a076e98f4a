for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra | python tools/ab_line.py drv-every5; done
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --time-every 0 | python tools/ab_line.py drv-notiming; done
