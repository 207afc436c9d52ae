# k_pfb_mr: workgroup size per channel count (MI355_PFB_MR_THREADS is read once per process)
for m in ${MS:-200 360 100 500 48}; do for th in 512 448 384 320 256; do
echo -n "M=$m TH=$th: "; MI355_PFB_MR_THREADS=$th timeout 100 python tools/r06_pfb100_probe.py $m 2>&1 | grep M= | sed 's/items=[0-9]* //; s/hbm_frac=//' | tr '\n' ' '; echo
done; done
