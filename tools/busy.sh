# GPU box: idle-gap analysis of the timed steps of one bench workload.  usage: busy.sh <tag> <bench args...>
tag=$1; shift
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -- python3 bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python3 tools/gpu_busy.py $(ls gpurun_out/prof_$tag/*/*.db | head -1) 3 3 > gpurun_out/${tag}_busy.txt 2>&1
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}_busy.txt
