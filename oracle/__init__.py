"""TEST INFRASTRUCTURE — CPU restatement (oracle) of the MCA/MMA fusion hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the reported CPU baseline.
The product path (``mca-paper_amd/``) never imports it and fails loudly when the HIP library is missing.

Pinned against the reference: ``oracle/make_goldens.py`` imports the reference implementation from
``/root/reference`` (in the build container only) and writes ``tests/golden/*.pt``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.
"""
