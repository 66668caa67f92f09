"""Prints the g++ flags for building csrc/torch_node.cpp against the installed PyTorch-ROCm (used by csrc/Makefile)."""
import os
import sysconfig

import torch

ti = os.path.join(os.path.dirname(torch.__file__), "include")
tl = os.path.join(os.path.dirname(torch.__file__), "lib")
print(" ".join([f"-I{ti}", f"-I{ti}/torch/csrc/api/include", "-I/opt/rocm/include", f"-I{sysconfig.get_paths()['include']}",
                "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
                f"-L{tl}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", f"-Wl,-rpath,{tl}",
                "-Wl,-rpath,/opt/rocm/lib", "-Wno-deprecated-declarations"]))
