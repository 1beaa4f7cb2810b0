R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
timeout 120 ./tools/micro/mfma_scale_probe > $O/mfma_scale_probe.txt 2>&1; echo "probe rc=$?" >> $O/mfma_scale_probe.txt
timeout 600 python tools/vendor_yardstick.py > $O/vendor_yardstick.txt 2>&1; echo "rc=$?" >> $O/vendor_yardstick.txt
timeout 300 python tools/inkernel_clock.py gemm > $O/inkernel_clock_gemm.txt 2>&1; echo "rc=$?" >> $O/inkernel_clock_gemm.txt
timeout 200 python tools/inkernel_clock.py attn > $O/inkernel_clock_attn.txt 2>&1; echo "rc=$?" >> $O/inkernel_clock_attn.txt
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -30 $O/mfma_scale_probe.txt; tail -20 $O/vendor_yardstick.txt; tail $O/inkernel_clock_gemm.txt $O/inkernel_clock_attn.txt; tail -c 600 $O/bench_default.json
