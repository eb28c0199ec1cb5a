# regenerate profiles/r06_* on the GPU box (outputs under gpurun_out/; copy the ones to keep into profiles/)
set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_cmd.sh r06 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/profile_r06.log 2>&1
bash tools/profile_cmd.sh r06_train python bench_train.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/profile_r06_train.log 2>&1
bash tools/profile_cmd.sh r06_soap python bench_soap.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/profile_r06_soap.log 2>&1
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
python bench_train.py > gpurun_out/r06_bench_train.json 2> gpurun_out/r06_bench_train.err
python bench_soap.py > gpurun_out/r06_bench_soap.json 2> gpurun_out/r06_bench_soap.err
python bench.py --boxes 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r06_bench_box1.json 2>/dev/null
python tools/gpu_md_probe.py 1000 3000 10000 2>/dev/null | grep "^{" > gpurun_out/r06_md_probe.json
# traffic files (on the build host, after copying the summaries into profiles/):
#   python tools/make_traffic_json.py profiles/r06_rocprofv3_summary.txt 1528404 profiles/r06_traffic.json
#   python tools/make_traffic_json.py profiles/r06_train_rocprofv3_summary.txt 1220632 profiles/r06_train_traffic.json "k_compress_h<f" "bench_train.py ..."
#   python tools/make_traffic_json.py profiles/r06_soap_rocprofv3_summary.txt 2617156 profiles/r06_soap_traffic.json k_soap_tail_fwd_set "bench_soap.py ..."
# one box over eight ranks, every rank's share timed one after the other on this GPU
python bench_pet_box.py --emulate-world 8 --steps 5 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/r06_box_emul8.json
python bench_pet_box.py --emulate-world 8 --exchange --steps 5 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/r06_box_emul8_exchange.json
python bench_soap.py --partition --emulate-world 8 --steps 5 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/r06_soap_box_emul8.json
