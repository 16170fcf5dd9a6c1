"""Process-wide render configuration (mirror of reference src/config.js:5-17).

The reference freezes `sampleRate` / `standardChunkSize` into prototypes and
wave tables when its modules load (selected with `--sampleRate=48000` on the
node command line).  Here they are plain module attributes: set them with
`dusp_amd.configure(sample_rate=48000)` BEFORE building a graph.
"""
standardChunkSize = 256  # the only chunk size the GPU path implements
sampleRate = 44100       # the reference's default (src/config.js:7)
