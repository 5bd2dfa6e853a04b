import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# sha256 of whole-file outputs, produced by the reference itself built from its own
# sources (oracle/Makefile, target ref) and reproduced by oracle/nw_oracle.c
HASHES = {
    "avpdb_csv": "2cde34c9d18d2081e95e3060ccc78726b8fc355070a1f17c3933b6963748af44",
    "avpdb_adjacent_out": "87c24224417491d193b8a544c21f4a0818dba45994506155c1c7de0bb6836cf2",
    "avpdb_triangular_out": "3eb3798aa4d2adff6aabe7bf69f06ef1425d2c25005b456f2a6440eff47e1e08",
    "mega_csv": "bc12a79ad0414d806acad1a4a846fba4e3ba26df1f15b7769161442dd5e70de6",
    "mega_adjacent_out": "667f3524a9ba31243de55c075165b336010ffeaf76600b39540b54f5dd085f18",
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hashes():
    return HASHES


@pytest.fixture(scope="session")
def avpdb_seqs():
    with open(os.path.join(GOLDEN, "avpdb.csv")) as f:
        return [line.split(",")[0] for line in f.read().split("\n")[1:] if line]


def sha256_file(path):
    import hashlib

    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()
