bash tools/gpu_r2_pmc.sh wg5 dec4.0,enc0.1,dec3.0 wgrad > /dev/null 2>&1
for f in sq1 sq2; do echo "## $f"; grep -h -A10 "wgrad_s1_v5" gpurun_out/wg5/pmc_$f.txt | head -12; done
grep -h -A1 "wgrad_s1_v5" gpurun_out/wg5/pmc_fetch.txt | head -3
grep -h "wgrad_s1_v5" gpurun_out/wg5/pmc_sq1.txt | tail -1
