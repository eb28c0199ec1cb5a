cd $GRAFT_REPO_ROOT
# round 5: the two-workgroups-per-CU kernels (k_emlp_s / k_emlp_bwd_s with the recomputed pre-activations, k_ablk_fwd4) and the
# VGPR-form adjoints forced on the small random batches of the parity fuzzer; then the default policy, the generic path, SOAP, training
for mode in default layernorm adaptive cosine species; do PET_FUZZ_FUSED=1 PET_FUZZ_SET=emlp_s=2 timeout 300 python tests/debug/fuzz_parity.py 51 25 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz5_forced_$mode.log; done
for mode in default legacy hypers; do timeout 300 python tests/debug/fuzz_parity.py 53 15 $mode 2>&1 | grep -v Warning | tail -8 > gpurun_out/fz5_parity_$mode.log; done
timeout 400 python tests/debug/fuzz_generic.py 51 12 2>&1 | grep -v Warning | tail -12 > gpurun_out/fz5_generic.log
timeout 300 python tests/debug/fuzz_soap.py 51 16 2>&1 | grep -v Warning | tail -10 > gpurun_out/fz5_soap.log
timeout 300 python tests/debug/fuzz_train.py 51 4 2>&1 | grep -v Warning | tail -6 > gpurun_out/fz5_train.log
tail -n 4 gpurun_out/fz5_*.log
