for cfg in "2048,3" "4096,3" "5904,3" "5904,2" "8000,3" "10000,3" "10000,2" "10000,4"; do
  echo -n "SPLIT=$cfg 10000 pairs: "; COATI_HIP_CK_SPLIT="$cfg" python tools/ab_fill.py --pairs 10000 --rounds 12 coati_amd/_build/libcoati_hip.so | cut -c34-100
done
for cfg in "2048,3" "8192,3" "16384,3" "35904,3" "40000,3"; do
  echo -n "SPLIT=$cfg 40000 pairs: "; COATI_HIP_CK_SPLIT="$cfg" python tools/ab_fill.py --pairs 40000 --rounds 8 coati_amd/_build/libcoati_hip.so | cut -c34-100
done
for cfg in "1904,3" "6000,3"; do
  echo -n "SPLIT=$cfg 6000 pairs: "; COATI_HIP_CK_SPLIT="$cfg" python tools/ab_fill.py --pairs 6000 --rounds 8 coati_amd/_build/libcoati_hip.so | cut -c34-100
done
