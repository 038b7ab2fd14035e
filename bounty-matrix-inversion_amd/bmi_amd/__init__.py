"""bmi_amd — host side of the MI355X-native TFHE engine behind the QFloat / qfloat_matrix_inverse API.

`tfhe`   : ctypes binding of libbmi_tfhe.so (include/bmi_tfhe.h), the C-ABI drop-in boundary.
Everything encrypted runs on the GPU through that library; there is no CPU fallback."""
