"""CPU oracle for the D-FINE training hot path -- TEST INFRASTRUCTURE ONLY.

A plain fp32 PyTorch/numpy/C restatement of the reference algorithms
(uc-vision/D-FINE-seg, /root/reference), each function citing the reference file:line it
follows. It is pinned against outputs of the reference itself: tests/golden/*.npz, produced
in the build container by oracle/gen_golden.py, which imports the reference's src/d_fine
(the reference ships no tests or golden vectors of its own for this path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (d-fine-seg_amd/) never does and has no CPU fallback.
"""
