for tp in 4 8 16; do for tpa in 8 16 32; do
echo "tp $tp tpa $tpa: $(EXP_AMD_THIN_TP=$tp EXP_AMD_THIN_TPA=$tpa python3 tools/bench_configs.py --only 4 --steps 40 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms_per_master_step']; n=d['kernel_scopes_per_master_step']
        print(round(d['ms_per_master_step'],3), {q:(k[q], n[q]) for q in k if 'thin' in q})
")"; done; done
