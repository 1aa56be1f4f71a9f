mkdir -p gpurun_out
for kd in "dense-row 0.25 14" "dense-column 0.25 14" "dense-row 1.4901161193847656e-08 8" "dense-column 1.4901161193847656e-08 8" "wide-window 0.25 6" "tiny 0.25 2" "square-ish 0.25 6" "duplicates-free-unsorted 0.25 6"; do
 echo "== $kd"; timeout -k 10 300 python tools/fixedit_probe.py $kd 2>&1 | grep "^k=" | cut -c1-230
done > gpurun_out/r4_c6b.log 2>&1; echo rc=$?; cat gpurun_out/r4_c6b.log
