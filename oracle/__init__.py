"""ORACLE - test infrastructure only (see oracle/refnet.py header). Never imported by pytorchcv_amd/."""
