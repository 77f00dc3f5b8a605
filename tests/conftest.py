import os
import sys

import pytest

# The GPU boxes of this pool show 256 logical CPUs but run under a cgroup quota of 16 cores (`/sys/fs/cgroup/cpu.max` = 1600000 100000).  torch's default of 128
# intra-op threads PER PROCESS -- the pytest process, every xdist worker, every spawned rank of the multi-process tests, each with its OpenMP workers
# spinning between parallel regions -- gets the whole job throttled: measured on twelve multi-rank tests, 202 s with the defaults against 32 s with four
# threads per process (`profiles/r04_gpu_suite_timing.md`).  Set before torch is imported anywhere, inherited by every child process.
for _k, _v in (("OMP_NUM_THREADS", "4"), ("MKL_NUM_THREADS", "4"), ("OMP_WAIT_POLICY", "PASSIVE")):
    os.environ.setdefault(_k, _v)
# Four xdist workers (and their spawned ranks) share ONE device: the engine's two memory-dependent switches (batched weight gradients, the kept SwiGLU product:
# engine.py `_mem_budget`) must not depend on what the neighbouring tests hold at that moment.  Under test they are decided from the device's total memory
# minus what THIS process holds; a test that then does not fit fails loudly (out of memory) instead of silently running another mode.
os.environ.setdefault("IE_MEM_BUDGET", "own")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def xport(port):
    """A rendezvous port private to this pytest-xdist worker (spawned ranks inherit PYTEST_XDIST_WORKER): tests that run at the same time are on
    different workers, so their process groups never meet on a port."""
    w = os.environ.get("PYTEST_XDIST_WORKER", "gw0")
    # (below the kernel's ephemeral range, 32768-60999: with `port + 1000 w` the fourth worker's ports sat inside it and a rendezvous once found its port taken by
    # a gloo pair connection -- EADDRINUSE in the round-6 suite; the tests' base ports lie in [29500, 30000))
    return 20000 + (int(port) - 29500) % 1000 + 1000 * (int(w[2:] or 0) % 12)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """Run the suite on four pytest-xdist workers by default (with the thread cap above: the whole GPU suite in ~100 s on one MI355X box, 860 s in round 3;
    the measurements behind it: profiles/r04_gpu_suite_timing.md).  `-n N` on the command line wins; IE_TEST_SERIAL=1 or a missing xdist plugin runs serially."""
    if (config.pluginmanager.hasplugin("xdist") and getattr(config.option, "numprocesses", None) is None and not os.environ.get("PYTEST_XDIST_WORKER")
            and os.environ.get("IE_TEST_SERIAL") != "1" and not getattr(config.option, "collectonly", False)):
        config.option.numprocesses = 4
        config.option.dist = "load"


# The longest tests first (eight ranks at 32 768 tokens / 7B width, the 8- and 4-rank layouts, the 7B-width engine runs), so that the short ones fill in
# around them instead of one of them forming the tail of the run.
_LONGEST_FIRST = ("test_isp_config3_layout_seq32768_sp8_at_7b_width", "test_sequence_parallel_sp4_sp8_equals_single_rank_step",
                  "test_llama2_tensor_parallel_2_with_hybrid_zero_on_8_ranks", "test_config3_shape_tensor2_weight4_on_internlm1_blocks_equals_one_rank",
                  "test_engine_7b_width_merged_benchmark_step_matches_oracle", "test_first_steps_of_the_benchmark_recipe_retrace_the_oracle_at_7b_width",
                  "test_engine_7b_shaped_layer_full_size_matches_oracle", "test_flash_attention_benchmark_regime_matches_oracle",
                  "test_weight_parallel_step_equals_resident_step", "test_moe_engine_expert_parallel_on_two_ranks_matches_the_reference_rules")


# ... and these LAST, when the other workers are running out of work: four staged ranks on one device (each imports torch and builds an engine) beside three
# busy workers hung once until its queue timeout (round 5); at the end of the run it has the box almost to itself.
_LAST = ("test_moe_engine_tensor_parallel_2_x_expert_parallel_2_on_four_ranks",)


def pytest_collection_modifyitems(config, items):
    rank = {n: i for i, n in enumerate(_LONGEST_FIRST)}
    rank.update({n: len(_LONGEST_FIRST) + 1 + i for i, n in enumerate(_LAST)})
    if os.environ.get("IE_TEST_FULL") != "1":   # parametrisations whose layout another test of the default run covers (each names it): IE_TEST_FULL=1 runs them too
        for it in items:
            if it.get_closest_marker("extended") is not None:
                it.add_marker(pytest.mark.skip(reason="extended parametrisation (its layout is covered by another test of the default run); IE_TEST_FULL=1 runs it"))
    items.sort(key=lambda it: rank.get(getattr(it, "originalname", None) or it.name, len(rank)))   # (stable: everything else keeps its file order)


def pytest_configure(config):
    if "torch" in sys.modules:   # (imported before this file by a plugin: the environment above came too late for this process)
        import torch

        torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ranks(n): GPUs a multi-rank test needs for its RCCL (nccl backend) variant")
    config.addinivalue_line("markers", "extended: a parametrisation whose layout another default test covers; run with IE_TEST_FULL=1")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
