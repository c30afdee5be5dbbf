"""The in-tree CLIP byte-pair encoder (lseg_hip.tokenizer.BPETokenizer, restating [3P] CLIP@04f4dc2
clip/simple_tokenizer.py) against an independent BPE implementation (HF `tokenizers`) on a synthetic merges file --
the real bpe_simple_vocab_16e6.txt.gz is not available offline, the algorithm is what is checked here: greedy
lowest-rank merges, the `</w>` end-of-word marker, lower-casing / whitespace collapsing, id layout
[256 bytes | 256 bytes+</w> | merges | SOT | EOT]."""
import gzip
import os

import pytest

from lseg_hip.tokenizer import BPETokenizer, _bytes_to_unicode

CORPUS = ("wall building sky floor tree ceiling road bed windowpane grass cabinet sidewalk person earth door table "
          "mountain plant curtain chair car water painting sofa shelf house sea mirror rug field armchair seat fence "
          "desk rock wardrobe lamp bathtub railing cushion base box column signboard chest counter sand sink "
          "skyscraper fireplace refrigerator grandstand path stairs runway case pool pillow screen stairway river "
          "bridge bookcase blind coffee toilet flower book hill bench countertop stove palm kitchen computer swivel "
          "boat bar arcade hovel bus towel light truck tower chandelier awning streetlight booth television airplane").split()


def _learn_merges(words, n):
    """Plain BPE training (most frequent adjacent pair first): gives a realistic rank table."""
    seqs = [tuple(w[:-1]) + (w[-1] + "</w>",) for w in words]
    merges = []
    for _ in range(n):
        cnt = {}
        for s in seqs:
            for a, b in zip(s, s[1:]):
                cnt[(a, b)] = cnt.get((a, b), 0) + 1
        if not cnt:
            break
        best = max(sorted(cnt), key=lambda p: cnt[p])
        merges.append(best)
        new = []
        for s in seqs:
            out, i = [], 0
            while i < len(s):
                if i < len(s) - 1 and (s[i], s[i + 1]) == best:
                    out.append(s[i] + s[i + 1]); i += 2
                else:
                    out.append(s[i]); i += 1
            new.append(tuple(out))
        seqs = new
    return merges


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    merges = _learn_merges(CORPUS, 120)
    path = os.path.join(tmp_path_factory.mktemp("bpe"), "vocab.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    return BPETokenizer(path), merges


def test_id_layout_and_specials(tok):
    t, merges = tok
    b2u = _bytes_to_unicode()
    assert t.enc[b2u[ord("a")]] < 256 and t.enc[b2u[ord("a")] + "</w>"] == t.enc[b2u[ord("a")]] + 256
    assert t.enc["".join(merges[0])] == 512
    assert t.enc["<|startoftext|>"] == 512 + len(t.ranks) and t.enc["<|endoftext|>"] == 513 + len(t.ranks)


def test_merges_match_an_independent_bpe(tok):
    tokenizers = pytest.importorskip("tokenizers")
    t, merges = tok
    vocab = dict(t.enc)
    model = tokenizers.models.BPE(vocab=vocab, merges=[tuple(m) for m in merges], end_of_word_suffix="</w>")
    hf = tokenizers.Tokenizer(model)
    for w in CORPUS + ["skywall", "treehouse", "zzz", "a", "windowpanes"]:
        ours = t.encode(w)
        theirs = hf.encode(w).ids
        assert ours == theirs, (w, ours, theirs)


def test_text_normalisation(tok):
    t, _ = tok
    assert t.encode("  Tree   HOUSE ") == t.encode("tree") + t.encode("house")     # lower-case, collapsed whitespace
    assert t.encode("tree&amp;house") == t.encode("tree") + t.encode("&") + t.encode("house")   # html.unescape
    assert t.encode("") == []


def test_decode_inverts_encode(tok):
    t, _ = tok
    for text in ("tree house", "windowpanes & sky", "a"):
        assert t.decode(t.encode(text)).strip() == text


# ---- real CLIP vocabulary (bpe_simple_vocab_16e6.txt.gz): runs where the file is available (LSEG_BPE_VOCAB or next to the package) ----
def _real_vocab():
    from lseg_hip import tokenizer as T
    return T._vocab_path()


@pytest.mark.skipif(_real_vocab() is None, reason="CLIP's bpe_simple_vocab_16e6.txt.gz is not on this machine (no network); set LSEG_BPE_VOCAB")
def test_real_clip_vocabulary_known_answers_and_the_150_ade_labels():
    """`clip.tokenize` (lseg_net.py:158,163-164) on the real vocabulary: published known-answer ids (CLIP / HF CLIPTokenizer documentation
    examples) and structural checks on the 150 ADE20K label strings the reference feeds it (lseg_module.py:97-109)."""
    import torch
    from lseg_hip.tokenizer import BPETokenizer, tokenize
    from lseg_hip.synth import read_labels
    tok = BPETokenizer(_real_vocab())
    SOT, EOT = 49406, 49407
    assert tok.enc["<|startoftext|>"] == SOT and tok.enc["<|endoftext|>"] == EOT and len(tok.enc) == 49408
    known = {"a photo of a cat": [320, 1125, 539, 320, 2368], "a diagram": [320, 22697], "a dog": [320, 1929], "a cat": [320, 2368]}
    for text, ids in known.items():
        assert tok.encode(text) == ids, (text, tok.encode(text))
    assert tok.encode("A  Photo\nof a CAT ") == known["a photo of a cat"]          # lower-casing, whitespace collapsing
    labels = read_labels(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lang-seg_amd", "label_files",
                                      "ade20k_objectInfo150.txt"))
    assert len(labels) == 150
    t = tokenize(labels, 77, 49408)
    assert t.shape == (150, 77) and t.dtype == torch.int64
    assert (t[:, 0] == SOT).all()
    eot = t.argmax(dim=-1)                                   # encode_text pools at the row maximum = the EOT position
    assert (t[torch.arange(150), eot] == EOT).all() and (eot >= 2).all() and (eot <= 8).all()
    for k in range(150):
        assert (t[k, eot[k] + 1:] == 0).all() and (t[k, 1:eot[k]] < SOT).all() and (t[k, 1:eot[k]] >= 0).all()
        assert tok.decode(t[k, 1:eot[k]].tolist()).strip() == labels[k].lower().strip(), (labels[k], t[k, :eot[k] + 1].tolist())
    assert t[labels.index("wall"), 1:3].tolist()[1] == EOT    # common single words are one token
