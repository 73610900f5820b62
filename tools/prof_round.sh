# end-of-round measurement pass (run through gpurun): headline bench + rocprof kernel stats, pix2pixHD patch / full-size benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r1k; rm -rf $O; mkdir -p $O
python bench.py --detail $O/detail.txt > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/stats.log 2>&1
python bench.py --model pix2pixHD --batch 32 --no_cpu_baseline --detail $O/p2p_patch_detail.txt > $O/p2p_patch_bench.json 2>$O/p2p_patch.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2p_stats -o run -- python bench.py --model pix2pixHD --batch 32 --steps 10 --warmup 3 --no_cpu_baseline > $O/p2p_stats.log 2>&1
python bench.py --model pix2pixHD --p2p_size 1024 --batch 2 --steps 5 --warmup 3 --no_cpu_baseline --detail $O/p2p_full_detail.txt > $O/p2p_full_bench.json 2>$O/p2p_full.err
python bench.py --infer > $O/infer.json 2>$O/infer.err
python bench.py --model sinskitG --netG resnet_9blocks --no_cpu_baseline > $O/resnet9.json 2>$O/resnet9.err
tail -c 300 $O/bench.json; ls $O/stats $O/p2p_stats
