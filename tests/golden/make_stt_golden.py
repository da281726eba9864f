#!/usr/bin/env python3
"""Generate tests/golden/stt_golden.npz from the reference's own fallback dependency for speech-to-text: transformers'
WhisperFeatureExtractor and WhisperForConditionalGeneration (reference: validation/stt/stt_validator.py:85-107), built from a
config - nothing is downloaded - with this repository's seeded weights.  Data only; needs transformers, not /root/reference.

    python tests/golden/make_stt_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import whisper as OW
    from rho_tts_amd import stt as S
    sys.path.insert(0, os.path.dirname(HERE))
    from test_oracle_whisper import clip
    cfg = S.tiny_test_config()
    x16 = OW.resample(clip(1.3, 24000, 3), 24000, 16000)
    mel = OW.log_mel(cfg, x16)
    model = OW.build(cfg, S.synthetic_state(cfg, 789))
    ids, first = OW.greedy(model, cfg, mel)
    c2 = S.SttConfig()
    mel2 = OW.log_mel(c2, OW.resample(clip(2.0, 24000, 4), 24000, 16000))
    np.savez_compressed(os.path.join(HERE, "stt_golden.npz"), tiny_pcm16k=x16, tiny_log_mel=mel.numpy(), tiny_ids=np.asarray(ids, np.int32),
                        tiny_first_logits=first.numpy(), tiny_enc=OW.encode(model, mel).numpy(), w_log_mel_head=mel2[:, :400].numpy())
    print("wrote stt_golden.npz:", len(ids), "token ids")



if __name__ == "__main__":
    main()
