"""ORACLE -- test infrastructure only.

CPU (torch fp32 + one small C file) restatement of the reference's RPN hot path
(SURVEY.md section 8a).  Nothing under ``nerf_rpn_amd/`` may import this package: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline.

Pinning: the reference is Python, so it is imported in the build container under the shims of
SURVEY.md App. C by ``tests/golden/make_golden.py``; that script compares every oracle function
with the reference on seeded inputs and writes the small ``tests/golden/*.npz`` fixtures that
travel to the GPU box.  ``sortv.c`` restates the reference's CUDA op (unbuildable here: needs
ATen/CUDA) and is pinned by known answers plus the reference's own Python IoU stack run on it.
"""
