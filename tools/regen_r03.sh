set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_r1.sh > gpurun_out/profile_r1.log 2>&1
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench_train.py > gpurun_out/r03_bench_train.json 2> gpurun_out/r03_bench_train.err
python bench_soap.py > gpurun_out/r03_bench_soap.json 2> gpurun_out/r03_bench_soap.err
python bench.py --boxes 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_bench_box1.json 2>/dev/null
python bench.py --total-boxes 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_strong_n1.json 2>/dev/null
bash tools/profile_train.sh > gpurun_out/profile_train.log 2>&1
python tools/gpu_md_probe.py 1000 3000 10000 2>/dev/null | grep "^{" > gpurun_out/r03_md_probe.json
# two ranks over gloo sharing the one GPU (functional checks of the N > 1 paths) and the one-box partitions
PET_BENCH_BACKEND=gloo python bench.py --gpus 2 --total-boxes 4 --boxes 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_bench_strong_gloo2.json
PET_BENCH_BACKEND=gloo python bench_train.py --gpus 2 --total-boxes 8 --micro 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03_bench_train_strong_gloo2.json
python bench_pet_box.py --emulate-world 8 --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r03_box_emul8.json
python bench_pet_box.py --emulate-world 8 --exchange --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r03_box_emul8_exchange.json
PET_BENCH_BACKEND=gloo python bench_pet_box.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r03_box_gloo2.json
PET_BENCH_BACKEND=gloo python bench_pet_box.py --gpus 2 --exchange --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r03_box_gloo2_exchange.json
