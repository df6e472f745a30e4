"""CPU oracle for the VQ codebook-lookup path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package (``vector_quantization_amd``) never imports it and has no CPU fallback.

* ``oracle.c_oracle``  — ctypes binding of ``vq_oracle.c`` (deterministic fp32 restatement; the index
  oracle: bit-exact target of the HIP path).
* ``oracle.torch_ref`` — the reference's composition of ATen ops restated line by line (float-tolerance
  reference on the GPU box and the timed CPU baseline); byte-identical to the reference on every fixture case.
* ``oracle.ref_import`` — BUILD CONTAINER ONLY: executes the reference's own quantizer-path source files from
  /root/reference behind a structure-only stand-in for the un-vendored ``todd``; ``oracle.make_golden`` generates
  every ``tests/golden/*.npz`` from it (``spec.source == 'reference-import'``) and ``tests/test_reference_pin.py``
  keeps the committed fixtures and ``torch_ref`` pinned to it.  Parity is therefore PINNED to the reference's
  files, except for three un-vendored todd definitions (``ema``, ``EMA``, ``MSELoss(norm=)``: SURVEY.md §8c).
"""
