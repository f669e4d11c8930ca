"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference (kkoutini/PaSST) hot path, used as the parity
checker by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg.  Nothing in ``passt_amd/`` (the product) imports this
package; the product path fails loudly when the HIP extension is missing.
"""
