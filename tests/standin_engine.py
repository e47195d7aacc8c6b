"""CPU stand-in for kyber_rs_amd.Engine: the methods bench.py's rank path calls, recording the calls and computing NOTHING.

Test infrastructure only (tests/test_multi_gpu_cpu.py, `bench.py --standin`): it lets the N > 1 plumbing — the self-spawning parent,
the rendezvous, the table broadcast, the shards, the max-over-ranks timing and the one JSON line — run end to end over gloo on a
host without a GPU.  It is not an engine: its outputs are never read, and the line bench.py prints with it says so."""
import numpy as np

BASE_TABLE_BYTES = 335232


class StandinEngine:
    def __init__(self, build_table=True):
        self.table = (np.arange(BASE_TABLE_BYTES, dtype=np.uint32) * 131 + 7).astype(np.uint8) if build_table else None
        self.calls = {}
        self.options = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # table image: what multi_gpu.distribute_base_table moves
    def base_table_export_dev(self, t):
        import torch
        t.copy_(torch.from_numpy(self.table))

    def base_table_import_dev(self, t):
        self.table = t.cpu().numpy().copy()

    def base_table(self):
        if self.table is None:
            raise RuntimeError("no table image was received")
        return self.table

    def sync(self, stream=0):
        pass

    def set_option(self, key, value):
        self.options[key] = int(value)

    def get_option(self, key):
        return self.options.get(key, 0)

    def profile_begin(self, n):
        pass

    def profile_read(self, cap=0):
        return []

    # the step closures of bench.py
    def mul_dev(self, scalars, **kw):
        self._count("mul_dev")

    def mul_base_dev(self, scalars, **kw):
        self._count("mul_base_dev")

    def sign_dev(self, *a, **kw):
        self._count("sign_dev")

    def verify_dev(self, *a, **kw):
        self._count("verify_dev")
