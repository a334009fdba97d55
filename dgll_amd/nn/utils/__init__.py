from .utils import accuracy, encode_onehot, load_data, normalize, sparse_mx_to_torch_sparse_tensor  # noqa: F401
