"""The resampler's lane assignment (lc3t_rs48_map in audio_codec_amd/csrc/lc3_enc_pre.inc, found by tools/rs48_map.py): a permutation of the 128 outputs,
both outputs of a lane of one filter phase, every half-wave's 32 input indices in 32 different LDS banks."""
import math, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rs48_map_is_a_conflict_free_permutation():
    src = open(os.path.join(ROOT, "audio_codec_amd", "csrc", "lc3_enc_pre.inc")).read()
    m = re.search(r"lc3t_rs48_map\[128\] = \{([^}]*)\}", src)
    v = [int(x) for x in m.group(1).replace("\n", " ").split(",")]
    assert len(v) == 128 and sorted(v) == list(range(128))
    n0, n1 = v[:64], v[64:]
    assert all(a % 4 == b % 4 for a, b in zip(n0, n1))                       # same polyphase branch: 15 n mod 4
    bank = lambda n: math.ceil(15 * n / 4) % 32                               # first input sample of output n (R/resamp12k8.c:48-57), LDS bank of a dword
    for g in (n0[:32], n0[32:], n1[:32], n1[32:]):
        assert len({bank(n) for n in g}) == 32


def test_rs48_lane4_table_is_the_generated_conflict_free_one():
    """lc3t_rs48_lane4 (lc3_enc_resample48_kernel): a permutation of (frame, b, p) codes whose sixteen lanes of every ds_read_b128 service group read sixteen
    different quad-words mod 16 - equal to what tools/rs48_map4.py builds"""
    import importlib.util
    src = open(os.path.join(ROOT, "audio_codec_amd", "csrc", "lc3_enc_pre.inc")).read()
    m = re.search(r"lc3t_rs48_lane4\[64\] = \{([^}]*)\}", src)
    v = [int(x) for x in m.group(1).replace("\n", " ").split(",")]
    spec = importlib.util.spec_from_file_location("rs48_map4", os.path.join(ROOT, "tools", "rs48_map4.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.conflict_free(v) and v == mod.build()


def test_b128_slot_table_makes_the_96_khz_resampler_conflict_free():
    """lc3t_b128_slot (lc3_enc_resample96_kernel_n*): lane -> group x 16 + place of the four ds_read_b128 service groups; with branch p = group & 1, frame
    (group >> 1) * FG + place / BB and b = place % BB a lane starts at quad-word (N / 4) f + 15 b + 2 p, and every group's sixteen lanes must differ mod 16 -
    for all three frame lengths, and the lanes of a wave must cover every (frame, b, p) of the step exactly once."""
    import importlib.util
    src = open(os.path.join(ROOT, "audio_codec_amd", "csrc", "lc3_enc_pre.inc")).read()
    m = re.search(r"lc3t_b128_slot\[64\] = \{([^}]*)\}", src)
    slot = [int(x) for x in m.group(1).replace("\n", " ").split(",")]
    assert sorted(slot) == list(range(64))
    spec = importlib.util.spec_from_file_location("rs48_map4", os.path.join(ROOT, "tools", "rs48_map4.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for g, lanes in enumerate(mod.GROUPS):
        assert sorted(slot[l] for l in lanes) == list(range(16 * g, 16 * g + 16))          # the table is the guide's grouping
    for NF in (960, 480, 240):
        LPF = NF // 30; FPI = 64 // LPF; BB = LPF // 2; FG = 16 // BB
        seen = set()
        for lanes in mod.GROUPS:
            q = set()
            for l in lanes:
                g, i = slot[l] >> 4, slot[l] & 15
                p, f, b = g & 1, (g >> 1) * FG + i // BB, i % BB
                assert f < FPI and (NF * f + 60 * b + 8 * p) % 4 == 0
                q.add(((NF * f + 60 * b + 8 * p) // 4) % 16)
                seen.add((f, b, p))
                assert NF * f + 60 * b + 8 * p + 168 <= 120 + 1920 + 8                    # the 42 reads stay inside the LDS image
            assert len(q) == 16, (NF, sorted(q))
        assert len(seen) == 64 and seen == {(f, b, p) for f in range(FPI) for b in range(BB) for p in range(2)}
