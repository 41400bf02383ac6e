# Same-box A/B of two builds of libgenie_smem.so (ab_old.so / ab_new.so at the repo root, built beforehand): the timing script
# runs alternately on each.  usage (GPU box): bash tools/ab_so.sh '<CASE>' [rounds]
R=$GRAFT_REPO_ROOT; C=${1:-100000,1000000,fromref,150}; N=${2:-3}
for i in $(seq 1 $N); do
  for v in old new; do
    cp $R/ab_$v.so $R/genie-smem_amd/libgenie_smem.so
    echo -n "$v: "; VARS="0" CASE=$C timeout -k 10 300 python $R/tools/experiments/r3q_queue.py 2>&1 | grep -v amdgpu.ids
  done
done
cp $R/ab_new.so $R/genie-smem_amd/libgenie_smem.so
