import os
import sys

os.environ.setdefault("MMFUSION_CONFIG_MKDIRS", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "simple-multimodal_amd")
for p in (REPO, PKG, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
