"""TEST INFRASTRUCTURE ONLY.

`oracle/` holds the CPU restatement of the reference's frame-generation hot path
(`restate.py`), the shim that imports the real reference from /root/reference when it is
present (`ref_shim.py`), and the golden-vector generator (`make_golden.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  Nothing under `monkey-net_amd/` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.
"""
