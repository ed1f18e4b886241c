export LNN_WGRAD_RING=0
bash tools/gpu_r2_pmc.sh wg4 dec4.0,enc0.1,dec3.0 wgrad > /dev/null 2>&1
for f in sq1 sq2; do echo "## $f"; grep -h -A11 "wgrad_s1_v4" gpurun_out/wg4/pmc_$f.txt | head -14; done
grep -h -A2 "wgrad_s1_v4" gpurun_out/wg4/pmc_fetch.txt | head -4
