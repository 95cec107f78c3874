# how many pairs to cut into row parts now that whole pairs keep tile-major checkpoints (cut pairs: step-major, stored through)
set -u
O=gpurun_out/r5cut4; mkdir -p $O
for rep in 1 2; do
timeout 900 python3 tools/split_ab.py 10000 4096,4 2048,4 3072,4 3584,4 4608,4 3072,3 4096,3 4096,5 3072,5 >> $O/n10000.txt 2>&1
done
