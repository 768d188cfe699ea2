set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_clk
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_clk -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --chunks 1 --steps 1 --warmup 1 > $R/gpurun_out/pmc_clk.log 2>&1 || { tail -20 $R/gpurun_out/pmc_clk.log; exit 1; }
cp $(find /tmp/pmc_clk -name '*counter_collection.csv') $R/gpurun_out/clk_counters.csv
cp $(find /tmp/pmc_clk -name '*kernel_trace.csv') $R/gpurun_out/clk_trace.csv
