set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh r3_single --no-two-level > gpurun_out/pmc_r3_single.log 2>&1; echo "single rc=$?"
bash tools/pmc_passes.sh r3_two --config two-level > gpurun_out/pmc_r3_two.log 2>&1; echo "two rc=$?"
python3 tools/pmc_two_level_traffic.py gpurun_out/pmc_r3_two gpurun_out/pmc_r3_two/pmc_traffic_two_level.json 8950216 > gpurun_out/pmc_r3_two/two_summary.txt 2>&1; echo "summary rc=$?"
tail -30 gpurun_out/pmc_r3_two/two_summary.txt
python3 -c "
import json; print(json.dumps(json.load(open('gpurun_out/pmc_r3_single/pmc_traffic.json'))['issue'], indent=1))"
