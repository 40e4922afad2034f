"""MI355X-native event-to-model photometric tracker for EDS (uzh-rpg/slam-eds hot path).

The directory name contains a hyphen, so load it with
``importlib.import_module("slam-eds_amd")`` (or ``import slam_eds_amd`` — the
alias module at the repo root).  Sub-modules:

* ``synth``   — deterministic synthetic workloads (numpy only)
* ``capi``    — ctypes binding of the C-ABI library ``csrc/libeds_hip.so``
* ``tracker`` — Python mirror of ``eds::tracking::Tracker`` (reference Tracker.hpp:36-114)
* ``batch``   — batched / multi-GPU alignment driver (one process per GPU, RCCL gather)
"""
__version__ = "0.1.0"
