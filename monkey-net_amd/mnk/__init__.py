"""Internal host-side glue of the MI355X-native Monkey-Net hot path: ctypes binding of libmonkeynet_hip.so
(`_lib`), autograd wrappers of the kernels (`ops`), SyncBN / gradient all-reduce helpers (`dist`)."""
