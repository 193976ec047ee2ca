"""What an fp16 hi pass + an fp8 lo pass (1.5 MFMA-units per product instead of the two bf16 passes) would cost in parity:
the full-width case of lo_map.py with every activation operand replaced by fp16(x) + the residual rounded to N mantissa bits (3 = e4m3),
emulated through the existing bf16 hi/lo pair.  Needs a one-off build of the dev library with the emulation compiled in:

    make -C ming_univision_amd/csrc clean && make -C ming_univision_amd/csrc -j8 all dev EXTRA=-DMN_EMUL_F16F8=3
    python tests/measure/f16f8_error.py            # on the GPU box
    make -C ming_univision_amd/csrc clean && make -C ming_univision_amd/csrc -j8 all dev      # back to the shipped dev library

The fp8 rounding of the WEIGHTS of the lo pass is not emulated (it adds an error term of the same size as the residual's rounding):
bracket it with N = 3 and N = 2.  Prints the `none` row of lo_map.py (all sites carry the emulated operand)."""
import os, runpy, sys
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "48", "no-site-has-this-prefix"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lo_map.py"), run_name="__main__")
