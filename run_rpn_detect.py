#!/usr/bin/env python
"""Drop-in entry point: same command line as the reference's nerf_rpn/run_rpn_detect.py, running on the MI355X HIP engine."""
from nerf_rpn_amd.run_rpn_detect import main

if __name__ == "__main__":
    main()
