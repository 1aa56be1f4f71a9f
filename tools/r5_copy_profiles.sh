# developer script (this container): what tools/r5_profiles.sh left under gpurun_out/r5prof -> profiles/r05_* (the traffic profile gets the git head)
O=gpurun_out/r5prof
for f in $O/bench_*.json; do cp $f profiles/r05_$(basename $f); done
cp $O/ks/k_kernel_stats.csv profiles/r05_kernel_stats_headline.csv
cp $O/ksf/k_kernel_stats.csv profiles/r05_kernel_stats_force_shard.csv
cp $O/dks/d_kernel_stats.csv profiles/r05_dense_kernel_stats.csv
cp $O/timeline.txt profiles/r05_timeline.txt
cp $O/timeline_force_shard.txt profiles/r05_timeline_force_shard.txt
cp $O/pmc_sq.txt profiles/r05_pmc_sq.txt
cp $O/pf/f_counter_collection.csv profiles/r05_pmc_fetch_counter_collection.csv
cp $O/pw/w_counter_collection.csv profiles/r05_pmc_write_counter_collection.csv
python - <<'PY'
import json, subprocess
d = json.load(open('gpurun_out/r5prof/pmc_traffic.json'))
d["git_head_when_post_processed"] = d.get("git_head_when_post_processed") or subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
json.dump(d, open('profiles/r05_pmc_traffic.json', 'w'), indent=1)
print('traffic profile: kernel sources', d['kernel_sources_sha16'], 'git', d['git_head_when_post_processed'])
import glob, os
for f in sorted(glob.glob('gpurun_out/r5prof/bench_*.json')):
    d = json.load(open(f)); r = d.get('roofline') or {}
    print(os.path.basename(f), d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'))
PY
head -12 profiles/r05_timeline.txt
