cd $GRAFT_REPO_ROOT
for mode in default adaptive grid legacy layernorm cosine hypers species; do timeout 300 python tests/debug/fuzz_parity.py 31 12 $mode 2>&1 | grep -v Warning | tail -14 > gpurun_out/fz_parity_$mode.log; done
for mode in residual cond cond-residual legacy; do timeout 300 python tests/debug/fuzz_mirror.py 31 8 $mode 2>&1 | grep -v Warning | tail -10 > gpurun_out/fz_mirror_$mode.log; done
timeout 300 python tests/debug/fuzz_train.py 31 6 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz_train.log
timeout 300 python tests/debug/fuzz_train.py 31 4 cond 2>&1 | grep -v Warning | tail -6 > gpurun_out/fz_train_cond.log
timeout 300 python tests/debug/fuzz_nl.py 31 30 2>&1 | grep -v Warning | tail -5 > gpurun_out/fz_nl.log
timeout 300 python tests/debug/fuzz_collate.py 31 10 2>&1 | grep -v Warning | tail -5 > gpurun_out/fz_collate.log
timeout 300 python tests/debug/fuzz_soap.py 31 12 2>&1 | grep -v Warning | tail -14 > gpurun_out/fz_soap.log
timeout 300 python tests/debug/fuzz_partition.py 31 6 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz_partition.log
timeout 300 python tests/debug/md_loop.py 2>&1 | grep -v Warning | tail -5 > gpurun_out/fz_md.log
