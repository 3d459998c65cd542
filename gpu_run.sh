export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -2
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1 -o bench -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/bench_prof.log 2>&1
cd /root/repo; tail -1 gpurun_out/bench_prof.log | cut -c1-300
ls -R gpurun_out/prof_r1 | head -20
