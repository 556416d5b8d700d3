"""The reference's evaluation harness flow (applications/eval_mtbench.py:267-386) on this package, with synthetic questions:
multi-turn questions answered through the `lade` drop-in surface (`lade.augment_all()`, `lade.config_lade(...)`, USE_LADE=1,
`model.generate`), per-turn timing, an answers .jsonl in the reference's record layout, the per-question stats file, the
AVERAGE THROUGHPUT line and the lade log (`lade.log_history()`, `lade.save_log(...)`).

There is no network, tokenizer or fastchat template on the build / test boxes: a "question" is a list of turns, each a list of
token ids, the conversation prompt of a turn is every earlier turn and answer followed by the new turn (what a chat template
produces, minus the role strings), and the answer record stores token ids where the reference stores decoded text.

  USE_LADE=1 python examples/eval_synthetic.py --answer-file /tmp/answers.jsonl --questions 4 --max-new-token 64
  USE_LADE=0 python examples/eval_synthetic.py ...            # the same harness on plain HF decoding
"""
import argparse
import json
import os
import random
import sys
import time
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

# temperature per category, as applications/eval_mtbench.py:37-46 configures it (greedy for the exact-answer categories)
temperature_config = {"writing": 0.7, "roleplay": 0.7, "extraction": 0.0, "math": 0.0, "coding": 0.0, "reasoning": 0.0, "stem": 0.1, "humanities": 0.1}


def synthetic_questions(n, vocab, seed=0):
    g = random.Random(seed)
    cats = sorted(temperature_config)
    qs = []
    for i in range(n):
        span = [g.randrange(3, vocab) for _ in range(20)]          # repeated spans, like natural text, give the n-gram pool something to find
        turns = [span + [g.randrange(3, vocab) for _ in range(6)] + span[:12], [g.randrange(3, vocab) for _ in range(8)] + span[4:16]]
        qs.append({"question_id": 81 + i, "category": cats[i % len(cats)], "turns": turns})
    return qs


def build_model(args):
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
    return model, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--answer-file", required=True)
    ap.add_argument("--model-id", default="synthetic-llama")
    ap.add_argument("--shape", default="tinyllama-1.1b")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16", "float32"])
    ap.add_argument("--questions", type=int, default=8)
    ap.add_argument("--num-choices", type=int, default=1)
    ap.add_argument("--max-new-token", type=int, default=128)
    ap.add_argument("--do-sample", type=int, default=1, help="0 forces greedy for every category (eval_mtbench.py:281)")
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--window", type=int, default=15)
    ap.add_argument("--guess", type=int, default=15)
    args = ap.parse_args()
    import lade
    lade.augment_all()
    lade.config_lade(LEVEL=args.level, WINDOW_SIZE=args.window, GUESS_SET_SIZE=args.guess, DEBUG=1, POOL_FROM_PROMPT=True)
    model, c = build_model(args)
    questions = synthetic_questions(args.questions, c["vocab"])
    answer_file = os.path.expanduser(args.answer_file)
    os.makedirs(os.path.dirname(answer_file) or ".", exist_ok=True)
    open(answer_file, "w").close()

    overall_time = overall_tp = overall_gen = count_gen = 0
    stats = {}
    for question_idx, question in enumerate(questions):
        temperature = temperature_config.get(question["category"], 0.7) if args.do_sample else 0.0
        stats[question_idx] = {}
        choices = []
        for i in range(args.num_choices):
            torch.manual_seed(i)
            random.seed(i)
            conversation, turns, prompts = [], [], []
            for j, qs in enumerate(question["turns"]):
                conversation = conversation + list(qs)
                prompts.append(list(conversation))
                input_ids = torch.tensor([conversation], device="cuda")
                do_sample = temperature >= 1e-4
                kw = dict(do_sample=True, temperature=temperature, top_k=0, top_p=1.0) if do_sample else dict(do_sample=False)
                torch.cuda.synchronize()
                start_time = time.time()
                output_ids = model.generate(input_ids, attention_mask=torch.ones_like(input_ids), max_new_tokens=args.max_new_token, **kw)
                torch.cuda.synchronize()
                gap_time = time.time() - start_time
                tokens = output_ids.numel() - input_ids.numel()
                overall_time += gap_time
                overall_gen += tokens
                overall_tp += tokens / gap_time
                count_gen += 1
                stats[question_idx][j] = [gap_time, tokens]
                print([f"step {i} turn {j} time: ", gap_time, " generated tokens: ", tokens, " throughput: ", tokens / gap_time])
                answer = output_ids[0, input_ids.numel():].tolist()
                turns.append(answer)
                conversation = conversation + answer
            choices.append({"index": i, "turns": turns, "prompts": prompts})
        with open(answer_file, "a") as fout:                     # one record per question, appended as it is answered
            fout.write(json.dumps({"question_id": question["question_id"], "answer_id": uuid.uuid4().hex[:22], "model_id": args.model_id,
                                   "choices": choices, "tstamp": time.time()}) + "\n")

    torch.save(stats[question_idx], answer_file + ".pt")
    print("LOG SAVE TO ", answer_file + ".pt")
    print(f"AVERAGE THROUGHPUT1 {overall_tp / count_gen} AVERAGE THROUGHPUT2 {overall_gen / overall_time} STAT {[overall_tp, count_gen, overall_gen, overall_time]}")
    lade.log_history()
    lade.save_log(answer_file + "-lade-log.pt")


if __name__ == "__main__":
    main()
