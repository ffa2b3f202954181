"""per-kernel GPU time of one whole-path step (rocprofv3 not needed): run under rocprofv3 --kernel-trace --stats"""
