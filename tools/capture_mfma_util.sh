# MFMA utilisation per kernel (north_star: "MFMA utilisation on the FC layers against gfx950 peak"): one PMC pass, kernel-trace only
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final5
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $R/gpurun_out/final5/pmc_mfma -- python $R/bench.py --steps 60 --warmup 70 --no-cpu-baseline --replay-size 100000 > $R/gpurun_out/final5/pmc_mfma.log 2>&1
ls $R/gpurun_out/final5/pmc_mfma/runc/
