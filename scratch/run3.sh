cd /tmp
for a in 0 1 2 3 4; do echo -n "abl=$a "; CGG_MSDA_BWD_ABL=$a python3 /root/repo/scratch/msda_bwd_only.py 0.5 5 2>&1 | grep host-level | sed 's/.*written)//'; done
for a in 0 1; do echo -n "c=4 abl=$a "; CGG_MSDA_BWD_C=4 CGG_MSDA_BWD_ABL=$a python3 /root/repo/scratch/msda_bwd_only.py 0.5 5 2>&1 | grep host-level | sed 's/.*written)//'; done
for r in 2 3; do echo -n "R=$r "; CGG_MSDA_BWD_R=$r python3 /root/repo/scratch/msda_bwd_only.py 0.5 5 2>&1 | grep host-level | sed 's/.*written)//'; done
