"""Only the vision topology of OpenAI CLIP that the B-cosified RN50 image encoder needs (`CLIP.clip.model`); the
tokenizer / text tower / weight download of the reference's vendored CLIP copy are outside the hot path."""
