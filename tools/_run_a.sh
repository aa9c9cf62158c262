timeout 300 python tools/halo_variants.py 2>&1 | grep -v Warning | tail -8
