# Round profile: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes.
# usage (on the GPU box): bash tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/kt_bench.json 2> $O/kt.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o pmc_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o pmc_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_write_bench.json 2> $O/pmc_write.err || exit 1
ls $O
