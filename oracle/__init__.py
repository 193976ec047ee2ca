"""oracle/ — CPU restatement of the reference algorithm for the hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import anything under this package, and only as the
checker.  The product package (ming_univision_amd) never imports it and fails
loudly when its HIP extension is missing.

Pinning status: the reference's own tests hold no golden vectors for synthetic
weights, and the real checkpoint is not available, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build
container by oracle/gen_golden.py (which imports /root/reference through
oracle/ref_shim.py) and committed as data under tests/golden/*.npz.
tests/test_oracle_golden.py checks every oracle function against them.
"""
