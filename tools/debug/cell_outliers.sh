# which batches of the round-5 sweeps sit above 1e-5 in dE/dR / dE/dcell, and what torch's own fp32 evaluation
# of the same model and inputs loses there (the yardstick, incl. dE/dcell).  bash tools/debug/cell_outliers.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
PET_FUZZ_FUSED=1 PET_FUZZ_SET=emlp_s=2 timeout 300 python tests/debug/fuzz_parity.py 51 25 adaptive 2>&1 | grep -v Warning | grep -E "ABOVE|worst" > gpurun_out/cell_outliers_adaptive.log
timeout 300 python tests/debug/fuzz_parity.py 53 15 hypers 2>&1 | grep -v Warning | grep -E "ABOVE|worst|num_gnn" > gpurun_out/cell_outliers_hypers.log
PET_FUZZ_FUSED=1 PET_FUZZ_SET=emlp_s=2 timeout 300 python tests/debug/fuzz_parity.py 51 25 species 2>&1 | grep -v Warning | grep -E "ABOVE|worst" > gpurun_out/cell_outliers_species.log
cat gpurun_out/cell_outliers_*.log
