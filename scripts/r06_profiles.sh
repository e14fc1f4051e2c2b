# the profile set of round 6 (run on the GPU box): stats + FETCH + WRITE passes per configuration, the sweep, the 8-rank
# tables, the host profiles and the long N-body run
tag=${1:-r06_p}
WITH_C5=1 scripts/profile_all.sh $tag > gpurun_out/${tag}_profile_all.log 2>&1; grep -c kernel gpurun_out/${tag}_profile_all.log
scripts/sweep.sh ${tag}_sweep
scripts/mr_kstats.sh ${tag}_mr8_512 --ranks 8 --mesh 512 --steps 12 --warmup 2 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_512_pencil --ranks 8 --mesh 512 --steps 12 --warmup 2 --np 2x4 --migrate 1 | tail -3
scripts/mr_kstats.sh ${tag}_mr8_1024_c4 --ranks 8 --mesh 1024 --steps 10 --warmup 2 | tail -3
mkdir -p gpurun_out/${tag}_host
python scripts/host_profile0.py 64 > gpurun_out/${tag}_host/host_profile0_64.txt 2>&1; head -3 gpurun_out/${tag}_host/host_profile0_64.txt
python scripts/host_profile0.py 64 bench > gpurun_out/${tag}_host/host_profile0_64_bench.txt 2>&1; head -3 gpurun_out/${tag}_host/host_profile0_64_bench.txt
PMESH_AMD_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29701 scripts/mp_host_profile.py 64 > gpurun_out/${tag}_host/mp_host_8.txt 2>&1; grep -m3 -i 'per cycle\|host' gpurun_out/${tag}_host/mp_host_8.txt
