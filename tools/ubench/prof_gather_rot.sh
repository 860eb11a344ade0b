#!/bin/bash
# rocprofv3 counter passes over tools/ubench/gather_rot (round 6): LDS conflict share, LDS / VALU busy, wait shares per kernel.
#   usage (GPU box): tools/ubench/prof_gather_rot.sh [blocks]   -> gpurun_out/r06_gather_rot_pmc.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
RAW=/tmp/grot_prof; rm -rf $RAW; mkdir -p $RAW gpurun_out
i=0
for CTRS in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-include-regex "direct_kernel|rot_kernel|par_kernel" --output-format csv -d $RAW/pmc$i -o pmc -- tools/ubench/gather_rot ${1:-1024} 20 > /dev/null 2> $RAW/pmc$i.err
done
python - <<'PY' > gpurun_out/r06_gather_rot_pmc.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob('/tmp/grot_prof/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name'].split('(')[0]][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for k in sorted(per):
    c = {n: sum(v.values()) / len(v) for n, v in per[k].items()}
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
    out = {'kernel': k, 'dispatches': len(next(iter(per[k].values())))}
    if c.get('SQ_LDS_IDX_ACTIVE'):
        out['lds_conflict_frac'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'], 4)
        out['lds_cycles_per_inst'] = round(c['SQ_LDS_IDX_ACTIVE'] / max(c.get('SQ_INSTS_LDS', 1), 1), 3)
        if cyc: out['lds_busy'] = round(c['SQ_LDS_IDX_ACTIVE'] / 256.0 / cyc, 3)
    if cyc:
        out['gpu_cycles'] = int(cyc)
        out['valu_insts_per_cu_cycle'] = round(c.get('SQ_INSTS_VALU', 0) / 256.0 / cyc, 3)
        out['valu_busy'] = round(c.get('SQ_ACTIVE_INST_VALU', 0) / 1024.0 / cyc, 3) if c.get('SQ_ACTIVE_INST_VALU') else None
    if c.get('SQ_WAVE_CYCLES'):
        out['wait_any'] = round(c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'], 3)
        out['wait_inst_lds'] = round(c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES'], 3)
    out['valu_per_lds_inst'] = round(c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_LDS', 1), 1), 2)
    out['salu_per_lds_inst'] = round(c.get('SQ_INSTS_SALU', 0) / max(c.get('SQ_INSTS_LDS', 1), 1), 2)
    print(out)
PY
tail -2 $RAW/*.err >> gpurun_out/r06_gather_rot_pmc.txt
cat gpurun_out/r06_gather_rot_pmc.txt
