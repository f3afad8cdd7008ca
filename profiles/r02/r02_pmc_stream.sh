cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${1:-r02g}; mkdir -p $O
bash profiles/pmc_pass.sh $O/pmc1 --steps 3 --warmup 1 || exit 1
bash profiles/pmc_pass3.sh $O/pmc3 --steps 3 --warmup 1 || exit 1
bash profiles/pmc_pass4.sh $O/pmc4 --steps 3 --warmup 1 || exit 1
for d in pmc1 pmc3 pmc4; do python profiles/pmc_summary.py $O/$d > $O/$d.txt 2>&1; done
grep -A30 "k_frame_stream" $O/pmc1.txt | head -34; grep -A14 "k_frame_stream" $O/pmc3.txt | head -16; grep -A14 "k_frame_stream" $O/pmc4.txt | head -16
