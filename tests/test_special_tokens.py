"""Loading of the delimiter-token dictionary (reference x2x_gen_unified.py:215-226 unpickles
special_tokens.pkl; this build reads it with a loader that executes nothing from the file)."""
import pickle

import pytest
import torch


def loader(tmp_path):
    from omnivideo.x2x_gen_unified import OmniVideoX2XUnified
    pipe = OmniVideoX2XUnified.__new__(OmniVideoX2XUnified)
    pipe.param_dtype, pipe.device = torch.bfloat16, torch.device("cpu")
    return lambda: pipe._load_special_tokens(str(tmp_path))


def tokens():
    g = torch.Generator().manual_seed(0)
    return {k: torch.randn(1, 32, generator=g) for k in ("<img_st>", "<img_ed>", "<prp_st>", "<prp_ed>")}


def test_absent_file_gives_none(tmp_path):
    assert loader(tmp_path)() is None


@pytest.mark.parametrize("name", ["special_tokens.pt", "special_tokens.pkl"])
def test_torch_saved_dictionary_loads(tmp_path, name):
    toks = tokens()
    torch.save(toks, tmp_path / name)
    got = loader(tmp_path)()
    assert set(got) == set(toks)
    for k in toks:
        assert got[k].dtype == torch.bfloat16 and torch.equal(got[k], toks[k].to(torch.bfloat16))


def test_raw_pickle_is_refused_with_the_fix(tmp_path):
    with open(tmp_path / "special_tokens.pkl", "wb") as f:
        pickle.dump(tokens(), f)
    with pytest.raises(RuntimeError, match="special_tokens.pt"):
        loader(tmp_path)()
    # the converted file is preferred over the raw pickle beside it
    torch.save(tokens(), tmp_path / "special_tokens.pt")
    assert set(loader(tmp_path)()) == set(tokens())
