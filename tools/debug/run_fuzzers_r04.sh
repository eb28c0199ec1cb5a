cd $GRAFT_REPO_ROOT
# round 4: the fused attention block forced on the small random batches, the generic path (fp32-MFMA GEMM, sliced attention), SOAP
for mode in default layernorm adaptive cosine species; do PET_FUZZ_FUSED=1 timeout 300 python tests/debug/fuzz_parity.py 41 25 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz4_fused_$mode.log; done
for mode in default legacy hypers; do timeout 300 python tests/debug/fuzz_parity.py 43 15 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz4_parity_$mode.log; done
timeout 400 python tests/debug/fuzz_generic.py 41 12 2>&1 | grep -v Warning | tail -12 > gpurun_out/fz4_generic.log
timeout 300 python tests/debug/fuzz_soap.py 41 16 2>&1 | grep -v Warning | tail -10 > gpurun_out/fz4_soap.log
timeout 300 python tests/debug/fuzz_train.py 41 4 2>&1 | grep -v Warning | tail -6 > gpurun_out/fz4_train.log
tail -n 4 gpurun_out/fz4_*.log
