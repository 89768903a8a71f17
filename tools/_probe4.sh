mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in blobs random; do
 rm -rf /tmp/p4; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o p4 -- python3 $R/tools/local_volume_bench.py --d 12 --labels $L > $R/gpurun_out/probe4_$L.log 2>&1 < /dev/null
 f=$(find /tmp/p4 -name "*kernel_stats.csv" | head -1); echo "== $L $f" >> $R/gpurun_out/probe4_stats.log
 if [ -n "$f" ]; then grep -E "local_fused|fill_f32|Name" "$f" | cut -c1-200 >> $R/gpurun_out/probe4_stats.log; fi
 grep "us per pair" $R/gpurun_out/probe4_$L.log
done
cat $R/gpurun_out/probe4_stats.log
