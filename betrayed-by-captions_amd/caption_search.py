"""Caption inference: beam search over the caption generator (open_set/utils/eval/inference.py:84-159, called from
open_set/models/mask2former_head.py:966-972 when `with_caption` / 'cap_results' is requested).

Same search as the reference, including its scoring quirks (they decide which sentence is returned):
  * the step logits are the MEAN over the decoder blocks of `generator(block_output)` (:92-93, :113);
  * candidates are ranked by (log p + parent weight) / length**alpha, the kept weight is de-normalised again (:118-120);
  * a continued sequence inherits `weights[row]` -- the new top-k array indexed by the PARENT's position (:141), not
    the candidate's own weight;
  * finished sequences are scored weight / len**alpha; the running maximum is reset at every step (:123-124), so the
    winner is the best sequence finished in the last step (else the first finished one); the search stops after
    `beam_width` finished sequences or when no live sequence is shorter than max_len - 1.
What differs is where it runs and how much of it: embeddings, transformer, generator, log-softmax and top-k stay on the
device and ONE small device->host copy per step (2 * beam_width numbers) drives the host bookkeeping (the reference moves
the full (beams, vocab) log-probabilities to the host every step); and the decoder runs INCREMENTALLY -- only the newest
position of every beam, against per-block key / value prefixes re-gathered by parent beam and cross-attention keys / values
computed once per image (`CaptionTransformer.begin_decode / decode_step`) -- where the reference re-runs every whole
sequence, the (beams, len, vocab) generator output included, at every step (:108-113). `kv_cache=False` keeps that
full re-run (the tests hold the two against each other).
"""
import torch


def _embed(head, ids):
    be = head.bert_embeddings
    return be.LayerNorm(be.word_embeddings(ids))


def beam_search(head, memory, BOS, EOS, max_len, beam_width=7, alpha=0.7, logging=False, tokenizer=None,
                return_ids=False, kv_cache=True):
    """memory (1, Q, d) = the image's query embeddings. Returns the decoded sentence (reference behaviour) or, with
    `return_ids` / when no tokenizer is available offline, the best token-id sequence (BOS ... EOS)."""
    if memory.shape[0] != 1:
        raise ValueError('beam_search decodes one image at a time (memory batch must be 1), as the reference does')
    dev = memory.device
    gen = head.caption_generator
    with torch.no_grad():
        tgt = _embed(head, torch.tensor([[BOS]], device=dev))                   # (1, 1, d)
        if kv_cache:
            state = gen.begin_decode(memory)
            outs = [o[0] for o in gen.decode_step(tgt, state)]
        else:
            outs = [o[0, 0, :] for o in gen(tgt=tgt, memory=memory)[0]]
        logits = torch.stack([gen.generator(o) for o in outs], 0).mean(0)
        logp = torch.log_softmax(logits[None, :].float(), dim=1)[0]
        w, cand = torch.topk(logp, k=beam_width, largest=True)
        weights, cand = w.cpu(), cand.cpu().tolist()
        seqs = [[BOS, c] for c in cand]
        parents = [0] * len(seqs)
        finished = []
        best_idx = 0
        keep = True
        while keep:
            # (sic) the reference re-initialises its running maximum at EVERY step (:123-124): the returned sentence is
            # the best one finished in the LAST step that finished any, or the first finished one overall
            best_score, best_idx = -100.0, 0
            ids = torch.tensor(seqs, dtype=torch.long, device=dev)             # (nb, len)
            nb, length = ids.shape
            if kv_cache:
                outs = gen.decode_step(_embed(head, ids[:, -1:]), state, torch.tensor(parents, dtype=torch.long, device=dev))
            else:
                outs = [o[:, -1, :] for o in gen(_embed(head, ids), memory.expand(nb, -1, -1).contiguous())[0]]
            logits = torch.stack([gen.generator(o) for o in outs], 0).mean(0)
            logp = torch.log_softmax(logits.float(), dim=1)                     # (nb, V)
            V = logp.shape[1]
            weighted = (logp + weights.to(dev)[:, None]) / length ** alpha
            w, pos = torch.topk(weighted.flatten(), k=min(beam_width, weighted.numel()), largest=True)
            w = (w * length ** alpha).cpu()                                     # de-normalised
            pos = pos.cpu().tolist()
            new_w, new_seqs, parents = [], [], []
            for idx, p in enumerate(pos):
                row, col = p // V, p % V
                seq = seqs[row] + [col]
                if col == EOS:
                    score = float(w[idx]) / len(seq) ** alpha
                    finished.append((seq, score))
                    if score > best_score:
                        best_score, best_idx = score, len(finished) - 1
                    if len(finished) == beam_width:
                        keep = False
                        break
                elif len(seq) < max_len - 1:
                    new_w.append(w[row])          # (sic) reference :141 indexes the new weights by the parent row
                    new_seqs.append(seq)
                    parents.append(row)
            if not new_seqs:
                keep = False
            else:
                weights = torch.stack(new_w) if new_w else weights
                seqs = new_seqs
    if logging:
        for s, sc in finished:
            print(s, sc)
    best = finished[best_idx][0] if finished else []
    if return_ids:
        return best
    if tokenizer is None:
        try:
            import transformers
            tokenizer = transformers.BertTokenizer.from_pretrained('bert-base-uncased', local_files_only=True)
        except Exception:
            return best
    res = ''
    for i, (s, _) in enumerate(finished):
        sentence = tokenizer.decode(s)
        if i == best_idx:
            res = sentence[1:-1]
    return res
