# Compile-time ablation instances of the eight-wave Winograd kernel (timing only, results are wrong): 1 no global loads after the first chunk,
# 8 no transform / LDS staging after the first chunk, 9 both.  bash tools/probes/wino_ablate.sh (through gpurun)
for a in 0 1 8 9; do echo "VTS_WINO_ABLATE=$a"; VTS_WINO_ABLATE=$a timeout 300 python tools/mb_wino.py 2>&1 | grep "N4 64->64 1024\|N4 256->256\|N4 512->512 128" | sed -e 's/:.*direct \([0-9]*\) us.*winograd \([0-9]*\) us.*/: direct \1 us, winograd \2 us/'; done
