"""CPU oracle for the VQ codebook-lookup path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package (``vector_quantization_amd``) never imports it and has no CPU fallback.

* ``oracle.c_oracle``  — ctypes binding of ``vq_oracle.c`` (deterministic fp32 restatement; the index
  oracle: bit-exact target of the HIP path).
* ``oracle.torch_ref`` — the reference's composition of ATen ops restated line by line (fixture
  generator, float-tolerance reference and the timed CPU baseline).
"""
