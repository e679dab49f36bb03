"""The documents cite evidence by path (profiles/..., tests/..., tools/..., csrc/...): every cited file must exist, so that a
renamed profile or a removed tool does not leave DESIGN.md / README.md / INTEGRATION.md pointing at nothing."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "acoustic_locating_vq-vae_amd")


def _cited(text):
    for m in re.finditer(r"`((?:profiles|tests|tools|oracle|include|csrc)/[A-Za-z0-9_./\-]+\.[A-Za-z0-9]+)", text):
        yield m.group(1)


def test_every_file_the_documents_cite_exists():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("tools", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(_cited(text)):
            if "*" in path or "rNN" in path or "r0N" in path:
                continue
            base = PKG if path.startswith("csrc/") else ROOT
            if not os.path.exists(os.path.join(base, path)):
                missing.append((doc, path))
    assert not missing, missing
