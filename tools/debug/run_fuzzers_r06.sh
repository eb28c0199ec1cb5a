cd $GRAFT_REPO_ROOT
# round 6: the same randomized sweeps as round 5 on the final code (the two-workgroup kernels and the fused attention block forced on
# the small random batches), SOAP-BPNN on the packed power spectrum (legacy) and on the full layout (Alchemical), the hyper-parameter
# sweep of SOAP-BPNN incl. the mlp head, the generic path, the training gradients
for mode in default layernorm adaptive cosine species; do PET_FUZZ_FUSED=1 PET_FUZZ_SET=emlp_s=2 timeout 300 python tests/debug/fuzz_parity.py 61 25 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz6_forced_$mode.log; done
for mode in default legacy hypers; do timeout 300 python tests/debug/fuzz_parity.py 63 15 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz6_parity_$mode.log; done
timeout 400 python tests/debug/fuzz_generic.py 61 12 2>&1 | grep -v Warning | tail -12 > gpurun_out/fz6_generic.log
timeout 400 python tests/debug/fuzz_soap.py 61 40 2>&1 | grep -v Warning | tail -10 > gpurun_out/fz6_soap.log
timeout 600 python tests/debug/fuzz_soap_hypers.py 61 30 2>&1 | grep -v Warning | tail -34 > gpurun_out/fz6_soap_hypers.log
timeout 300 python tests/debug/fuzz_train.py 61 4 2>&1 | grep -v Warning | tail -6 > gpurun_out/fz6_train.log
# the same with the shared-ring row kernels forced on the small random batches (so_rows_s.hip serves every generic GEMM of the second-order pass)
PET_FUZZ_SET=emlp_s=2 timeout 300 python tests/debug/fuzz_train.py 61 4 2>&1 | grep -v Warning | tail -6 > gpurun_out/fz6_train_forced.log
for f in gpurun_out/fz6_*.log; do echo "== $f"; tail -n 3 $f; echo "   batches flagged above 1e-5 (any of the three): $(grep -c ABOVE $f)"; done > gpurun_out/r06_fuzz_summary.txt
cat gpurun_out/r06_fuzz_summary.txt
