#!/bin/bash
# Counter passes over k_oracle_net (the fused TractOracle-Net kernel), 4 096 rows per launch:
# one rocprofv3 --pmc pass per counter group (never combined with a trace domain other than
# the kernel trace).    bash profiles/collect_pmc_oracle_net.sh r04   (from the repo root)
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out/pmc_oracle_net
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16" \
             "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES" \
             "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/g$i -- \
      python3 $root/benchmarks/bench_oracle_net.py 4096 > $out/g$i.json 2> $out/g$i.log
done
python3 - <<PY > $root/profiles/${tag}_pmc_oracle_net.txt
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob('$out/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_oracle_net' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('k_oracle_net<4>, 4096 streamlines per launch; counter sums per launch (mean over launches)')
for k in sorted(tot):
    print(f'{k:36s} {tot[k] / cnt[k]:16.0f}   ({cnt[k]} launches)')
PY
cat $root/profiles/${tag}_pmc_oracle_net.txt
