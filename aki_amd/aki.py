"""AKI model class: host-side mirror of src/aki.py (train-time AKI) with the same constructor and
forward signature; see also src/modeling_aki.py:83-151 (HF-Hub twin, same forward).

Reference: /root/reference/codes/open_flamingo/src/aki.py  (__init__ :10-50, set_trainable :52-57, forward :65-134)
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from .helpers import PerceiverResampler
from .vlm import VLMWithLanguageStream


def sample_next(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, generator=None) -> torch.Tensor:
    """One sampling step on [B, V] logits, in HF's processor order: temperature, top-k, top-p (nucleus), multinomial."""
    x = logits.float() / temperature
    V = x.shape[-1]
    if 0 < top_k < V:
        kth = x.topk(top_k, dim=-1).values[:, -1:]
        x = x.masked_fill(x < kth, float("-inf"))
    if top_p < 1.0:
        sx, si = x.sort(dim=-1, descending=True)
        cum = sx.softmax(dim=-1).cumsum(dim=-1)
        drop = cum - sx.softmax(dim=-1) >= top_p            # keep the smallest prefix whose mass reaches top_p (always >= 1 token)
        sx = sx.masked_fill(drop, float("-inf"))
        x = torch.full_like(x, float("-inf")).scatter(-1, si, sx)
    return torch.multinomial(x.softmax(dim=-1), 1, generator=generator).squeeze(-1)


def _embedding_tables(emb: nn.Module, logits: torch.Tensor):
    """(weight, additional weight or None, max_original_id) when ops.greedy_pick can gather the picked token's embedding row itself
    (bf16 tables that hold a row for every logit column), else None: the module's own forward is used."""
    w = getattr(emb, "weight", None)
    if w is None or w.dtype != torch.bfloat16 or not w.is_contiguous() or w.device != logits.device or w.shape[1] % 8:
        return None
    n_add = int(getattr(emb, "num_additional_embeddings", 0) or 0)
    if n_add > 0:
        extra, max_orig = emb.additional_embedding.weight, int(emb.max_original_id)
        if extra.dtype != torch.bfloat16 or not extra.is_contiguous() or extra.device != w.device or w.shape[0] <= max_orig:
            return None
        rows = max_orig + 1 + extra.shape[0]
    elif type(emb) is nn.Embedding or hasattr(emb, "max_original_id"):
        extra, max_orig, rows = None, w.shape[0] - 1, w.shape[0]
    else:
        return None
    return (w, extra, max_orig) if logits.shape[-1] <= rows else None


class AKI(VLMWithLanguageStream):
    def __init__(self, vision_encoder: nn.Module, lang_model: nn.Module, vis_feature_dim: int, initial_tokenizer_len: int,
                 pad_token_id: int, decoder_layers_attr_name: str = None, gradient_checkpointing: bool = False,
                 base_img_size: Optional[int] = None, num_vision_tokens: int = 144):
        self._special_tokens = {"media_token": "<image>", "end_of_trunk_token": "<|endofchunk|>"}
        lang_embedding_dim = lang_model.get_input_embeddings().weight.shape[1]
        super().__init__(
            vision_encoder=vision_encoder,
            vision_tokenizer=PerceiverResampler(dim=vis_feature_dim, dim_inner=lang_embedding_dim, num_latents=num_vision_tokens),
            lang_model=lang_model, initial_tokenizer_len=initial_tokenizer_len, gradient_checkpointing=gradient_checkpointing,
            base_img_size=base_img_size, decoder_layers_attr_name=decoder_layers_attr_name, pad_token_id=pad_token_id)

    def set_trainable(self):
        """Unfreeze everything except the vision_encoder (src/aki.py:52-57)."""
        self.requires_grad_(True)
        self.vision_encoder.requires_grad_(False)

    def _should_apply_weight_decay(self, parameter_name):
        return True

    def default_eos_token_ids(self):
        """What `lang_model.generate` would stop on by default: generation_config.eos_token_id (Phi-3.5-mini-instruct ships
        [32007, 32001, 32000]) when the checkpoint carried one, else config.eos_token_id."""
        for holder in (getattr(self.lang_model, "generation_config", None), getattr(self.lang_model, "config", None)):
            eos = getattr(holder, "eos_token_id", None) if holder is not None else None
            if eos is not None:
                return [int(eos)] if isinstance(eos, int) else [int(e) for e in eos]
        return []

    def forward(self, vision_x: Optional[torch.Tensor], lang_x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, image_size: Optional[Tuple] = None,
                past_key_values: Optional[List[Union[torch.Tensor, Tuple[torch.Tensor]]]] = None,
                past_media_locations: Optional[torch.Tensor] = None, past_vision_tokens: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = False, **kwargs):
        """vision_x (B, T_img, F=1, C, H, W); lang_x (B, T_txt) with <image> placeholders -> CausalLMOutputWithPast
        whose logits cover the EXPANDED stream length L (src/aki.py:65-134)."""
        assert not (past_vision_tokens is None) ^ (past_media_locations is None), \
            "past_vision_tokens and past_media_locations must both be None or both be not None"
        if vision_x is not None:
            plan = self._start_splice_plan(lang_x) if past_key_values is None else None   # ahead of the vision side's launches
            vision_tokens = self.vision_tokenizer(self._encode_vision_x(vision_x=vision_x))
        else:
            vision_tokens, plan = None, None
        new_inputs = self._prepare_inputs_for_forward(
            vision_tokens=vision_tokens, lang_x=lang_x, attention_mask=attention_mask, vision_attention_mask=None,
            labels=labels, past_key_values=past_key_values, past_media_locations=past_media_locations,
            padding_side="right", past_vision_tokens=past_vision_tokens, splice_plan=plan)
        output = self.lang_model(**new_inputs, use_cache=use_cache, past_key_values=past_key_values, **kwargs)
        self._post_forward_hook()
        return output

    def _beam_search(self, cache, logits, K: int, max_new_tokens: int, eos_ids, pad_id: int, length_penalty: float, early_stopping):
        """Beam search over the decode path (the algorithm of HF `GenerationMixin` beam search with a `BeamSearchScorer`:
        2K candidates per step, hypotheses normalised by generated_length ** length_penalty, one sequence returned per sample).
        The prompt's cache rows are expanded to K beams and re-ordered in place every step (AkiKVCache.select_rows)."""
        dev = logits.device
        B = logits.shape[0]
        if max_new_tokens <= 0:                        # nothing to generate: the greedy path returns an empty tensor too
            return torch.zeros((B, 0), dtype=torch.long, device=dev)
        cache.select_rows(torch.arange(B, device=dev).repeat_interleave(K))
        logp = torch.log_softmax(logits.float(), dim=-1).repeat_interleave(K, dim=0)             # [B*K, V]
        V = logp.shape[-1]
        scores = torch.zeros((B, K), dtype=torch.float32, device=dev)
        scores[:, 1:] = -1e9                                                                      # all beams start identical: keep one
        seqs = torch.zeros((B * K, 0), dtype=torch.long, device=dev)
        hyps = [[] for _ in range(B)]                                                             # per sample: (score, tokens)
        done = [False] * B
        eos_set = set(int(e) for e in eos_ids)

        def worst(b):
            return min(h[0] for h in hyps[b]) if len(hyps[b]) >= K else -float("inf")

        def add(b, score_sum, toks, gen_len):
            sc = score_sum / (max(gen_len, 1) ** length_penalty)
            if len(hyps[b]) < K or sc > worst(b):
                hyps[b].append((sc, toks))
                if len(hyps[b]) > K:
                    hyps[b].remove(min(hyps[b], key=lambda h: h[0]))

        for t in range(max_new_tokens):
            cand = (logp.view(B, K, V) + scores[:, :, None]).view(B, K * V)
            top_s, top_i = cand.topk(2 * K, dim=-1)
            top_s_h, top_i_h = top_s.tolist(), top_i.tolist()
            nxt_tok = torch.full((B, K), pad_id, dtype=torch.long)
            nxt_beam = torch.zeros((B, K), dtype=torch.long)
            nxt_score = torch.full((B, K), -1e9, dtype=torch.float32)
            seqs_h = seqs.tolist() if eos_set else None
            for b in range(B):
                if done[b]:
                    nxt_beam[b] = torch.arange(K)
                    continue
                n = 0
                for rank, (sc, idx) in enumerate(zip(top_s_h[b], top_i_h[b])):
                    beam, tok = divmod(idx, V)
                    if tok in eos_set:
                        if rank < K:                                                              # HF: EOS beyond the K best is ignored
                            add(b, sc, seqs_h[b * K + beam] + [tok], t + 1)
                        continue
                    nxt_tok[b, n], nxt_beam[b, n], nxt_score[b, n] = tok, beam, sc
                    n += 1
                    if n == K:
                        break
                if len(hyps[b]) >= K:
                    best_running = float(nxt_score[b].max()) / ((t + 1) ** length_penalty)
                    if early_stopping is True or (early_stopping is False and worst(b) >= best_running):
                        done[b] = True
            if all(done) or t + 1 == max_new_tokens:
                # close the books: running beams become hypotheses (with the token chosen at this step)
                for b in range(B):
                    if done[b]:
                        continue
                    for n in range(K):
                        if float(nxt_score[b, n]) > -1e8:
                            base = seqs[b * K + int(nxt_beam[b, n])].tolist()
                            add(b, float(nxt_score[b, n]), base + [int(nxt_tok[b, n])], t + 1)
                break
            gather = (torch.arange(B)[:, None] * K + nxt_beam).view(-1).to(dev)
            seqs = torch.cat([seqs.index_select(0, gather), nxt_tok.view(-1, 1).to(dev)], dim=1)
            scores = nxt_score.to(dev)
            cache.select_rows(gather)
            step_logits = self.lang_model.decode_step(input_ids=nxt_tok.view(-1).to(dev), past_key_values=cache)
            logp = torch.log_softmax(step_logits.float(), dim=-1)
        best = [max(h, key=lambda x: x[0])[1] for h in hyps]
        width = max(len(x) for x in best)
        out = torch.full((B, width), pad_id, dtype=torch.long)
        for b, x in enumerate(best):
            out[b, : len(x)] = torch.tensor(x, dtype=torch.long)
        return out.to(dev)

    @torch.no_grad()
    def generate(self, vision_x, lang_x, image_size=None, attention_mask=None, past_key_values=None,
                 past_media_locations=None, past_vision_tokens=None, **kwargs):
        """Generation (src/aki.py:136-209 + src/aki_generation.py:36-86; local_demo.py / eval.py call it with
        do_sample=False): MMA prefill into a KV cache, then one HIP decode step per token.  Like HF `generate` called with
        `inputs_embeds` only, the return value holds just the NEW tokens [B, <= max_new_tokens]; finished rows are padded
        with pad_token_id.  Decoding modes, selected by the HF keyword arguments the reference forwards (`**kwargs`,
        src/aki.py:160-207): greedy (default), sampling (`do_sample=True` with `temperature`, `top_k`, `top_p`, optional
        `generator`), beam search (`num_beams=K`, `length_penalty`, `early_stopping`; one returned sequence per sample).
        Differences from the reference, both only visible for B > 1 (where the reference is inconsistent): the
        prompt batch is right-padded and every sample continues from its own length."""
        num_beams = int(kwargs.pop("num_beams", 1))
        do_sample = bool(kwargs.pop("do_sample", False))
        temperature = float(kwargs.pop("temperature", 1.0))
        top_k = int(kwargs.pop("top_k", 0) or 0)
        top_p = float(kwargs.pop("top_p", 1.0))
        rng = kwargs.pop("generator", None)
        length_penalty = float(kwargs.pop("length_penalty", 1.0))
        early_stopping = kwargs.pop("early_stopping", False)
        if int(kwargs.pop("num_return_sequences", 1)) != 1:
            raise NotImplementedError("num_return_sequences > 1")
        if num_beams < 1 or (num_beams > 1 and do_sample):
            raise NotImplementedError("beam-sample decoding (num_beams > 1 with do_sample=True)")
        if do_sample and temperature <= 0:
            raise ValueError("temperature must be positive")
        if past_key_values is not None:
            raise NotImplementedError("generate() starts from a fresh prefill")
        max_new_tokens = int(kwargs.pop("max_new_tokens", kwargs.pop("max_length", 20)))
        # HF `generate` stops on generation_config.eos_token_id when the caller passes none (the reference's callers pass only
        # max_new_tokens / do_sample: local_demo.py:76-87, eval_cv_bench/eval.py:99-104); `eos_token_id=[]` switches it off.
        eos = kwargs.pop("eos_token_id", None)
        if eos is None:
            eos = self.default_eos_token_ids()
        eos_ids = set([eos] if isinstance(eos, int) else (eos or []))
        pad_id = kwargs.pop("pad_token_id", self.pad_token_id)
        use_graph = kwargs.pop("use_graph", None)
        if use_graph is None:
            use_graph = max_new_tokens >= 8          # capture costs about two eager steps
        if vision_x is None:
            raise NotImplementedError("text-only generation is outside the AKI hot path")
        plan = self._start_splice_plan(lang_x)
        vision_tokens = self.vision_tokenizer(self._encode_vision_x(vision_x=vision_x))
        new_inputs = self._prepare_inputs_for_forward(vision_tokens=vision_tokens, lang_x=lang_x, attention_mask=attention_mask,
                                                      padding_side="right", splice_plan=plan)
        table = new_inputs["attention_mask"]
        L = new_inputs["inputs_embeds"].shape[1]
        out = self.lang_model(inputs_embeds=new_inputs["inputs_embeds"], attention_mask=table, use_cache=True,
                              cache_capacity=L + max_new_tokens, last_token_logits=True)
        cache = out.past_key_values
        B = lang_x.shape[0]
        logits = out.logits[:, 0]                                                  # logits of each sample's last real token
        if num_beams > 1:
            tokens = self._beam_search(cache, logits, num_beams, max_new_tokens, eos_ids, pad_id, length_penalty, early_stopping)
            self._post_forward_hook()
            return tokens
        tokens = torch.full((B, max_new_tokens), pad_id, dtype=torch.long, device=lang_x.device)
        done = torch.zeros(B, dtype=torch.bool, device=lang_x.device)
        eos_t = torch.tensor(sorted(eos_ids), dtype=torch.long, device=lang_x.device) if eos_ids else None
        lm = self.lang_model
        # The one-launch decode chain (one sequence, decode_chain.hip) bounds every dependency wait; a wait that gives up leaves garbage in that
        # step's output and a sticky error word.  Both loops below read the word wherever they synchronise anyway (every 8th token) and once at the
        # end; on an error the tokens after the last verified point are decoded again on the five-launch-per-layer path: nothing unverified is
        # ever returned (lm.decode_verified switches the chain off for this cache and warns).
        chained = lambda: getattr(cache, "chain", None) is not None
        stepper = None
        if use_graph and not do_sample and logits.is_cuda and logits.dtype == torch.bfloat16:
            # Greedy: the pick (argmax, pad for finished rows, append, eos check, cache_len advance, the next step's embedding row) is one launch
            # behind the decode step (inside the replayed graph where there is one); the host looks at the finished flags every 8th token
            # instead of syncing per token.
            from . import ops
            from .phi3 import DecodeGraph
            done8 = torch.zeros(B, dtype=torch.uint8, device=lang_x.device)
            done_at = torch.full((B,), -1, dtype=torch.int32, device=lang_x.device)
            start_len, host_len0 = cache.cache_len.clone(), cache.host_len
            pick = dict(pad_token_id=pad_id, eos_ids=eos_t, done=done8, tokens=tokens, start_len=start_len, done_at=done_at)
            ids = torch.zeros(B, dtype=torch.long, device=lang_x.device)
            emb_mod = lm.get_input_embeddings()
            embed = _embedding_tables(emb_mod, logits)
            nxt_emb = None if embed is None else torch.empty((B, emb_mod.weight.shape[1]), dtype=torch.bfloat16, device=lang_x.device)
            pick_e = pick if embed is None else dict(pick, embed=embed, next_embeds=nxt_emb)
            ops.greedy_pick(logits.contiguous(), ids, cache_len=cache.cache_len, advance=False, **pick_e)      # token 0, from the prefill
            t, t_ok = 1, 1                                  # tokens[:, :t_ok] are verified (token 0 comes from the prefill, not from the chain)
            period = 8 if eos_t is not None else 32         # host synchronisations: the EOS check needs them often, the chain's verification alone does not
            while True:
                ok = True
                while t < max_new_tokens:
                    if t % period == 0 and (eos_t is not None or chained()):
                        if chained() and not lm.decode_verified(cache):
                            ok = False
                            break
                        t_ok = t
                        if eos_t is not None and bool(done8.all()):
                            break
                    if stepper is not None:
                        stepper.step_greedy()
                    else:
                        # One sequence on the one-launch decode chain is three launches per token (chain, head, pick + embedding):
                        # the host runs far ahead of them and a hipGraph would only add its capture (about 12 ms per call, 7 tokens' worth).
                        # Anything else - batches, the five-launch-per-layer path - is ~165 launches per token and is captured after its first
                        # eager step.
                        if nxt_emb is not None:
                            lg = lm.decode_step(inputs_embeds=nxt_emb, past_key_values=cache, advance=False)
                        else:
                            lg = lm.decode_step(input_ids=ids, past_key_values=cache, advance=False)
                        ops.greedy_pick(lg, ids, cache_len=cache.cache_len, advance=True, **pick_e)
                        if not chained():
                            stepper = DecodeGraph(lm, cache, greedy=pick)
                            stepper.ids.copy_(ids)
                    t += 1
                if ok and chained() and not lm.decode_verified(cache):
                    ok = False
                if ok:
                    break
                # recovery: back to the last verified token, then on without the chain.  The finished flags are rebuilt from the verified tokens
                # (a batch can hold rows that finished before that point; one sequence would have ended the loop there)
                t = t_ok
                cache.cache_len.copy_(start_len + (t_ok - 1))
                cache.host_len = host_len0 + (t_ok - 1)
                tokens[:, t_ok:] = pad_id
                done8.zero_()
                done_at.fill_(-1)
                if eos_t is not None:
                    hit = (tokens[:, :t_ok, None] == eos_t[None, None, :]).any(-1)
                    fin = hit.any(1)
                    done8.copy_(fin.to(torch.uint8))
                    done_at.copy_(torch.where(fin, hit.to(torch.int32).argmax(1).to(torch.int32), torch.full_like(done_at, -1)))
                ids.copy_(tokens[:, t_ok - 1])
                if nxt_emb is not None:
                    nxt_emb.copy_(emb_mod(ids))
            steps = t
            if eos_t is not None and bool(done8.all()):
                steps = int(done_at.max()) + 1
            self._post_forward_hook()
            return tokens[:, :steps]
        if use_graph:
            from .phi3 import DecodeGraph
            stepper = DecodeGraph(lm, cache)
        t, n_out = 0, max_new_tokens
        ck = None                                           # the state at the last verified point of a chained decode
        while True:
            ok = True
            while t < max_new_tokens:
                if t % 8 == 0 and (t == 0 or chained()):
                    if chained() and not lm.decode_verified(cache):
                        ok = False
                        break
                    ck = (t, logits.clone(), done.clone(), cache.cache_len.clone(), cache.host_len)
                nxt = sample_next(logits, temperature, top_k, top_p, rng) if do_sample else logits.float().argmax(dim=-1)
                nxt = torch.where(done, torch.full_like(nxt, pad_id), nxt)
                tokens[:, t] = nxt
                t += 1
                if eos_t is not None:
                    done = done | (nxt[:, None] == eos_t[None, :]).any(-1)
                    if bool(done.all()):
                        n_out = t
                        break
                if t < max_new_tokens:
                    logits = stepper.step(nxt) if stepper is not None else lm.decode_step(input_ids=nxt, past_key_values=cache)
            if ok and chained() and not lm.decode_verified(cache):
                ok = False
            if ok:
                break
            t, logits, done, cl, cache.host_len = ck
            cache.cache_len.copy_(cl)
            tokens[:, t:] = pad_id
            n_out, stepper = max_new_tokens, None           # a graph captured around the chain is gone with it
            if use_graph:
                stepper = DecodeGraph(lm, cache)
        self._post_forward_hook()
        return tokens[:, :n_out]
