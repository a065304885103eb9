"""The file rendezvous of the C-ABI communicator (pt_comm_create_from_file_tagged, csrc/pt_comm.hip) — the parts that need neither a GPU
nor RCCL: a waiting rank must NOT take the id of another job (stale file, other world size, other job tag) and must give up after its
time limit; world 1 never touches the file.  (world > 1 itself has not run on real devices yet: DESIGN.md section 6.)"""
import ctypes as C
import os
import struct
import tempfile
import time

import pytest

import ptamd

MAGIC = 0x50544944


def _write(path, world, tag, magic=MAGIC):
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQ", magic, world, tag) + bytes(range(128)))


@pytest.mark.parametrize("content", ["missing", "stale_tag", "other_world", "bad_magic", "short"])
def test_waiting_rank_ignores_files_of_other_jobs(content):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "job.id")
        if content == "stale_tag":
            _write(path, 2, 41)
        elif content == "other_world":
            _write(path, 4, 42)
        elif content == "bad_magic":
            _write(path, 2, 42, magic=0x12345678)
        elif content == "short":
            open(path, "wb").write(b"\0" * 100)         # e.g. the 128-byte file format of round 2 cut short
        h = C.c_void_p()
        t0 = time.time()
        rc = ptamd.lib().pt_comm_create_from_file_tagged(path.encode(), 42, 1, 2, 0, 1, C.byref(h))
        assert rc != 0 and not h.value
        assert 0.9 < time.time() - t0 < 5.0                     # it waited its second, then gave up
        msg = ptamd.lib().pt_last_error().decode()
        assert "timed out" in msg and "job tag 42" in msg
        if content != "missing":
            assert os.path.exists(path)                         # a reader never removes anything


def test_world_one_needs_no_file_and_no_rccl():
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "never_written.id")
        c = ptamd.Comm(rank=0, world=1, device=0, id_file=path, job_tag=7)
        assert c.rank == 0 and c.world == 1 and not os.path.exists(path)
        c.close()
    with pytest.raises(ptamd.PtError):
        ptamd.Comm(rank=2, world=2, device=0, id_file="/nonexistent/x", job_tag=1, timeout_s=0)
