cd $GRAFT_REPO_ROOT
timeout 3000 bash tools/collect_profiles.sh r5 > gpurun_out/collect_r5.log 2>&1
tail -5 gpurun_out/collect_r5.log
ls gpurun_out/r5 | head -60
cat gpurun_out/r5/bench_bench_plain.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for k,v in d['supplementary_hbm_regime'].items():
    r=v.get('roofline',{})
    print(k[:70], v.get('ms'), r.get('frac'), r.get('frac_survey_8d'), r.get('frac_of_fill'))
print(json.dumps(d.get('other_workloads'))[:600])
print(d.get('step_breakdown'))
"
