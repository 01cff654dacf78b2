"""The committed counter files that bench.py's `roofline.traffic` / `hbm_frac` are read from belong to the build that is benched:
`profiles/r06_pmc_gather2_items.json` carries the commit in which the sources of the 2-hop gather last changed when its PMC passes ran
(scripts/pmc_gather2.sh); if those sources change again, the file must be regenerated (measurement 4b)."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

SOURCES = ["ggad_amd/csrc/hop2_ldsw.hip", "ggad_amd/csrc/plan_build.cpp", "ggad_amd/csrc/plan.hip"]


def test_gather_pmc_file_is_stamped_with_the_last_change_of_the_gather_sources():
    path = os.path.join(ROOT, "profiles", "r06_pmc_gather2_items.json")
    assert os.path.exists(path), "run scripts/pmc_gather2.sh on a GPU box and commit its JSON under profiles/"
    doc = json.load(open(path))
    for key in ("20", "150"):
        e = doc["by_batches_per_launch"][key]
        assert e["hbm_bytes_per_neighbour"] and 10.0 < e["hbm_bytes_per_neighbour"] < 200.0
        assert e["k_tile_counts_hbm_bytes_per_neighbour"] and e["k_tile_counts_hbm_bytes_per_neighbour"] > 0
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (a snapshot on the GPU box): the stamp is checked where the repository is")
    try:
        out = subprocess.run(["git", "log", "-1", "--format=%h", "--"] + SOURCES, cwd=ROOT, check=True, capture_output=True, text=True)
        dirty = subprocess.run(["git", "status", "--porcelain", "--"] + SOURCES, cwd=ROOT, check=True, capture_output=True, text=True)
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("git not usable here")
    head = out.stdout.strip()
    assert not dirty.stdout.strip(), "uncommitted changes in the gather sources: commit them, then regenerate the PMC file"
    assert head and (doc["commit"].startswith(head) or head.startswith(doc["commit"])), \
        f"profiles/r06_pmc_gather2_items.json was measured on {doc['commit']}, the gather sources last changed in {head}: regenerate it"


def test_chunk_kernel_pmc_file_is_stamped_with_the_last_change_of_its_sources():
    """bench.py's `by_kernel` entry of k_train_chunk_xcd takes its issue floor from profiles/r05_pmc_chunk_xcd.json (the entry
    rested on a literal): the file carries the commit in which the chunk kernel's sources last changed when its PMC passes ran
    (scripts/pmc_chunk_xcd.sh)."""
    path = os.path.join(ROOT, "profiles", "r05_pmc_chunk_xcd.json")
    assert os.path.exists(path), "run scripts/pmc_chunk_xcd.sh on a GPU box and commit its JSON under profiles/"
    doc = json.load(open(path))
    assert 1.0 < doc["issue_us_per_step"] < doc["kernel_us_per_step"] < 200.0
    c = doc["counters"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES"):
        assert c.get(k, 0) > 0, k
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here (a snapshot on the GPU box): the stamp is checked where the repository is")
    src = ["ggad_amd/csrc/step_xcd.hip", "ggad_amd/csrc/step_common.h"]
    try:
        out = subprocess.run(["git", "log", "-1", "--format=%h", "--"] + src, cwd=ROOT, check=True, capture_output=True, text=True)
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("git not usable here")
    head = out.stdout.strip()
    assert head and (doc["commit"].startswith(head) or head.startswith(doc["commit"])), \
        f"profiles/r05_pmc_chunk_xcd.json was measured on {doc['commit']}, the chunk kernel's sources last changed in {head}: regenerate it"
