# PMC passes of the two-level step with the batch as one part, for the lockstep and the lanes form of the top-level kernel
set -u
cd $GRAFT_REPO_ROOT
for V in 1 2; do
  bash tools/pmc_passes.sh r3_top$V --config two-level --top-variant $V --batch-parts 1 > gpurun_out/pmc_r3_top$V.log 2>&1; echo "variant $V rc=$?"
  PARTS=1 python3 tools/pmc_two_level_traffic.py gpurun_out/pmc_r3_top$V gpurun_out/pmc_r3_top$V/two.json 8950216 > gpurun_out/pmc_r3_top$V/two_summary.txt 2>&1
  python3 -c "
import json; j=json.load(open('gpurun_out/pmc_r3_top$V/two.json')); print({k:v for k,v in j['issue'].items() if k.startswith('topTraverse')}); print(j['by_kernel'])"
done
