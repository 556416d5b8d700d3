"""The reference's `minimal.py` flow on this package: optional `lade.augment_all()` +
`lade.config_lade(...)`, one warm-up `generate`, then a timed sampling run and a timed greedy run, printed in the
reference's format.

  python examples/minimal.py                              # plain HF decoding
  LOAD_LADE=1 USE_LADE=1 python examples/minimal.py       # lookahead decoding on the HIP step engine

There is no network on the build / test boxes, so by default the model is a random-weight Llama of the named shape
(`--shape tinyllama-1.1b`, fp16) and the prompt is synthetic token ids; with `--model <path>` a local HF checkpoint and
its tokenizer are used exactly as in the reference.  With LOAD_LADE=1 the greedy output is also compared id-for-id
with plain HF greedy decoding of the same model (`USE_LADE=0`)."""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def build_model(args):
    if args.model:
        from transformers import AutoModelForCausalLM, AutoTokenizer
        tok = AutoTokenizer.from_pretrained(args.model)
        model = AutoModelForCausalLM.from_pretrained(args.model, torch_dtype=torch.float16, device_map="cuda")
        model.tokenizer = tok
        text = ("<|system|>\nYou are a friendly chatbot who always responds in the style of a pirate.</s>\n<|user|>\n"
                "How do you fine tune a large language model?</s>\n<|assistant|>")
        return model, tok, tok(text, return_tensors="pt").to("cuda")
    from transformers import LlamaConfig, LlamaForCausalLM
    from lookaheaddecoding_amd.weights import make_config
    c = make_config(args.shape, max_pos=4096)
    if args.layers:
        c["layers"] = args.layers
    cfg = LlamaConfig(vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["layers"],
                      num_attention_heads=c["heads"], num_key_value_heads=c["kv_heads"], max_position_embeddings=c["max_pos"],
                      rms_norm_eps=c["eps"], tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=None)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = LlamaForCausalLM(cfg).to({"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[args.dtype]).eval()
    g = torch.Generator().manual_seed(123)
    # a prompt with repeated spans, like natural text: gives the n-gram pool something to find
    span = torch.randint(3, c["vocab"], (24,), generator=g)
    ids = torch.cat([span, torch.randint(3, c["vocab"], (8,), generator=g), span[:16]]).unsqueeze(0).cuda()
    return model, None, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=None, help="local HF checkpoint directory (default: random weights of --shape)")
    ap.add_argument("--shape", default="tinyllama-1.1b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16", "float32"],
                    help="random-weight logits are nearly flat: in 16-bit the greedy ids of two different attention implementations "
                         "diverge at the first near-tie; float32 shows the id-for-id identity (tests/test_gpu_hf.py asserts it)")
    ap.add_argument("--max-new-tokens", type=int, default=256)
    ap.add_argument("--level", type=int, default=7)
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--guess", type=int, default=20)
    args = ap.parse_args()
    assert torch.cuda.is_available()
    load_lade = bool(int(os.environ.get("LOAD_LADE", 0)))
    if load_lade:
        import lade
        lade.augment_all()
        lade.config_lade(LEVEL=args.level, WINDOW_SIZE=args.window, GUESS_SET_SIZE=args.guess, DEBUG=1, POOL_FROM_PROMPT=True)
    model, tok, inputs = build_model(args)
    n_in = inputs["input_ids"].numel()
    new = args.max_new_tokens

    random.seed(1)
    model.generate(**inputs, max_new_tokens=1, do_sample=False)                      # warm up
    torch.cuda.synchronize()
    t0s = time.time()
    torch.manual_seed(0)
    sample_output = model.generate(**inputs, max_new_tokens=new, do_sample=True, temperature=0.7, top_k=50, top_p=0.9)
    torch.cuda.synchronize()
    t1s = time.time()
    t0g = time.time()
    greedy_output = model.generate(**inputs, max_new_tokens=new, do_sample=False)
    torch.cuda.synchronize()
    t1g = time.time()

    print("\nOutput:\n" + 100 * "-")
    if tok is not None:
        print("Greedy output: ", tok.decode(greedy_output[0], skip_special_tokens=False))
        print("Sample output: ", tok.decode(sample_output[0], skip_special_tokens=False))
    else:
        print("Greedy output ids: ", greedy_output[0, n_in:n_in + 32].tolist(), "...")
        print("Sample output ids: ", sample_output[0, n_in:n_in + 32].tolist(), "...")
    print("Greedy Generated Tokens:", greedy_output.numel() - n_in, "Generation Speed: ", (greedy_output.numel() - n_in) / (t1g - t0g), " tokens/s")
    print("Sample Generated Tokens:", sample_output.numel() - n_in, "Generation Speed: ", (sample_output.numel() - n_in) / (t1s - t0s), " tokens/s")
    if load_lade:
        lade.log_history()
        if int(os.environ.get("USE_LADE", 0)):
            os.environ["USE_LADE"] = "0"                                              # read per generate call (lade/decoding.py:16)
            t0 = time.time()
            plain = model.generate(**inputs, max_new_tokens=new, do_sample=False)
            torch.cuda.synchronize()
            t1 = time.time()
            os.environ["USE_LADE"] = "1"
            same = plain.shape == greedy_output.shape and bool((plain == greedy_output).all())
            n_same = int((plain[0, :greedy_output.shape[1]] == greedy_output[0, :plain.shape[1]]).long().cumprod(0).sum()) - n_in
            print("Plain HF greedy of the same model:", (plain.numel() - n_in) / (t1 - t0), " tokens/s;  lookahead greedy ids identical:", same,
                  f"(first {n_same} of {new} new tokens agree)")


if __name__ == "__main__":
    main()
