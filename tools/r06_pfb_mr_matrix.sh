# k_pfb_mr: which phase costs what (MI355_PFB_MR_DBG: 1 no branch-filter arithmetic, 2 no transform passes, 4 no input loads, 8 no stores)
for cfg in ${CFGS:-"8 0" "8 256" "8 512"}; do set -- $cfg
for dbg in ${DBGS:-0 1 2 4 8 15}; do
echo -n "FS=$1 TH=$2 DBG=$dbg: "; MI355_PFB_MR_FS=$1 MI355_PFB_MR_THREADS=$2 MI355_PFB_MR_DBG=$dbg timeout 100 python tools/r06_pfb100_probe.py ${MS:-100 20 48 200 360} 2>&1 | grep M= | sed 's/items=[0-9]* //; s/hbm_frac=//' | tr '\n' ' '; echo
done; done
