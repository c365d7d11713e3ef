"""Special token ids and result-key mapping of the captioning path.
Same values as the reference's config/Constants.py:1-18 (they are part of the
checkpoint / vocabulary contract); the data/checkpoint base paths and the POS
tag table of the corpus tools are out of scope here."""
PAD = 0
UNK = 1
BOS = 2
EOS = 3
MASK = 4
VIS = 5

PAD_WORD, UNK_WORD, BOS_WORD = '<pad>', '<unk>', '<bos>'
EOS_WORD, MASK_WORD, VIS_WORD = '<eos>', '<mask>', '<vis>'

# criterion name -> (prediction key, ground-truth key) in the results dict
mapping = {
    'lang': ('tgt_word_logprobs', 'tgt_word_labels'),
    'length': ('pred_length', 'tgt_length'),
}
