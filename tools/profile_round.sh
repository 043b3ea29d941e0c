# The rocprofv3 runs behind profiles/rNN_* (run on the GPU box from the repo root; outputs under gpurun_out/prof_<tag>/).
# usage: bash tools/profile_round.sh <tag>
set -e
tag=${1:-r}
out=$PWD/gpurun_out/prof_$tag
repo=$PWD
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o b -- python3 $repo/bench.py --no-cpu-baseline > $out/bench_rocprof_run.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -o s -- python3 $repo/bench.py --no-cpu-baseline --in-flight 1 --steps 40 > $out/serial_rocprof_run.json 2> $out/serial.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $repo/tools/kbench.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $repo/tools/kbench.py > $out/write.log 2>&1
cd $repo
find $out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | while read f; do cp $f $out/$(basename $(dirname $(dirname $f)))_$(basename $f); done
rm -rf $out/bench $out/serial $out/fetch $out/write
ls -la $out
