cd /root/repo
for w in minified utf8 pretty4; do
  bash scripts/prof.sh v15_$w --workload $w > gpurun_out/prof_v15_$w.txt 2>&1
  tail -40 gpurun_out/prof_v15_$w.txt | cut -c1-220
done
