"""vognet-pytorch_amd — MI355X-native VOGNet forward path.

Python surface = the reference's plugin boundary (`mdl_selector.get_mdl_loss_eval`,
`AnetBaseMdl(cfg, comm)`, `forward(dict) -> dict`, `main_dist` CLI); compute =
hand-written HIP kernels in `csrc/` behind the C ABI of `include/vog_hip.h`.
The directory name contains a hyphen: import it with
`importlib.import_module("vognet-pytorch_amd")` or through the `vognet_amd`
alias module at the repo root.
"""
__version__ = "0.1.0"
