"""The step record as a poll: a spin that runs out because a step was merely slow must not degrade the decoder for the rest of
its life; only a record that is still absent from the mapped buffer after the stream has drained turns polling off."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_e2e import load, make_engine


def _run(stall):
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    run = load("e2e_greedy.json")["runs"][0]
    cfg, w, eng = make_engine(run, torch.float32)
    dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]), use_graph=True)
    dec.start(run["prompt"], run["eos"], random.Random(run["seed"]))
    calls = {"timed_out": 0}
    real = dec.st.poll_record

    def flaky(step_no, timeout_s=None):
        if stall == "transient" and timeout_s is None and step_no % 3 == 0:
            calls["timed_out"] += 1
            return None                           # the bounded spin ran out (host descheduled, profiler attached ...)
        if stall == "absent":
            calls["timed_out"] += 1
            return None                           # the device's stores never reach the mapped buffer
        return real(step_no, timeout_s)

    dec.st.poll_record = flaky
    while len(dec.tokens) < run["max_length"] and not dec.finished_by_eos:
        dec.step()
    return run, dec, calls


def test_a_slow_step_keeps_polling_and_an_absent_record_turns_it_off():
    run, dec, calls = _run("transient")
    assert calls["timed_out"] >= 2 and dec.poll is True
    assert dec.tokens[:run["max_length"]] == run["tokens"]
    run, dec, calls = _run("absent")
    assert dec.poll is False and calls["timed_out"] >= 1
    assert dec.tokens[:run["max_length"]] == run["tokens"]
