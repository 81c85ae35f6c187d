for cfg in 256x256x32 256x128x64 512x128x32 256x128x32; do echo "== $cfg"; DFW_BIG_CFG=$cfg timeout -k 10 120 python scratch/bench_conv.py vae512,dec512,vae256,vae128 2>&1 | grep -v amdgpu.ids; done
